import numpy as np
import pytest
import torch

from eventgpt_amd.constants import EVENT_TOKEN_INDEX, IGNORE_INDEX
from eventgpt_amd.data import (ByteTokenizer, EventImageProcessor, event_frames,
                               prepare_event_prompt, synthetic_events, tokenizer_event_token)
from eventgpt_amd.models import (EventChatConfig, EventChatModel, get_spatio_temporal_features,
                                 kv_cache_bytes)


@pytest.fixture(scope="module")
def tiny():
    torch.manual_seed(0)
    cfg = EventChatConfig.tiny()
    return EventChatModel.from_config(cfg, torch.float32, "cpu", seed=0)


def _frames(n=5, size=28):
    proc = EventImageProcessor(size)
    return [proc(f, return_tensors="pt")["pixel_values"][0]
            for f in event_frames(synthetic_events(3000, 64, 48), n)]


def test_pooling_shapes():
    assert get_spatio_temporal_features(torch.randn(5, 577, 16)).shape == (582, 16)
    assert get_spatio_temporal_features([torch.randn(577, 16)]).shape == (578, 16)
    f = torch.randn(3, 7, 4)
    out = get_spatio_temporal_features(f, num_temporal_tokens=5)
    assert out.shape == (12, 4)
    assert torch.allclose(out[:3], f.mean(1)) and torch.all(out[3:5] == 0)
    assert torch.allclose(out[5:], f.mean(0))
    assert get_spatio_temporal_features(f, num_temporal_tokens=2).shape == (9, 4)
    with pytest.raises(ValueError):
        get_spatio_temporal_features(torch.randn(4, 4))


def test_kv_cache_size_matches_published_figure():
    cfg = EventChatConfig.eventgpt_7b()
    assert kv_cache_bytes(cfg, 1) == 512 * 1024          # 0.5 MB per token
    assert abs(kv_cache_bytes(cfg, 636) / 2**20 - 318.0) < 0.01


def test_splice_positions_and_labels(tiny):
    ids = torch.tensor([[1, 10, EVENT_TOKEN_INDEX, 11, 12]])
    labels = torch.tensor([[1, 10, IGNORE_INDEX, 11, 12]])
    feats = torch.randn(1, 6, tiny.config.hidden_size)
    mask = torch.ones_like(ids)
    pos = torch.arange(5)[None]
    out = tiny.prepare_inputs_labels_for_multimodal(ids, pos, mask, None, labels,
                                                    event_features=feats, already_adapted=True)
    new_ids, new_pos, new_mask, past, embeds, new_labels = out
    assert new_ids is None and past is None
    assert embeds.shape == (1, 10, tiny.config.hidden_size)
    emb = tiny.model.embed_tokens
    assert torch.allclose(embeds[0, :2], emb(ids[0, :2]))
    assert torch.allclose(embeds[0, 2:8], feats[0])
    assert torch.allclose(embeds[0, 8:], emb(ids[0, 3:]))
    assert new_labels[0].tolist() == [1, 10] + [IGNORE_INDEX] * 6 + [11, 12]
    assert new_mask.shape == (1, 10) and new_mask.dtype == mask.dtype and bool(new_mask.all())
    assert new_pos[0].tolist() == list(range(10))
    # None in -> None out
    out2 = tiny.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                     event_features=feats, already_adapted=True)
    assert out2[1] is None and out2[2] is None and out2[5] is None


def test_cached_features_double_adaptor_semantics(tiny):
    ids = torch.tensor([[1, EVENT_TOKEN_INDEX, 5]])
    feats = torch.randn(1, 4, tiny.config.hidden_size)
    e_ref = tiny.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                      event_features=feats)[4]
    e_fix = tiny.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                      event_features=feats, already_adapted=True)[4]
    fa = tiny.get_model().feature_adaptor
    assert torch.allclose(e_ref[0, 1:5], fa(feats[0]), atol=1e-5)
    assert torch.allclose(e_fix[0, 1:5], feats[0])


def test_decode_step_and_no_event_passthrough(tiny):
    ids = torch.tensor([[7]])
    out = tiny.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                    event_features=torch.zeros(1, 2, 64))
    assert out[0] is ids and out[4] is None
    out = tiny.prepare_inputs_labels_for_multimodal(torch.tensor([[1, 2]]), None, None, "kv", None)
    assert out[3] == "kv" and out[4] is None


def test_batch_right_padding(tiny):
    ids = torch.tensor([[1, EVENT_TOKEN_INDEX, 5, 6], [1, 9, EVENT_TOKEN_INDEX, 0]])
    mask = torch.tensor([[1, 1, 1, 1], [1, 1, 1, 0]])
    feats = [torch.randn(3, 64), torch.randn(5, 64)]
    _, _, new_mask, _, embeds, _ = tiny.prepare_inputs_labels_for_multimodal(
        ids, None, mask, None, None, event_features=feats, already_adapted=True)
    assert embeds.shape == (2, 7, 64)
    assert new_mask.tolist() == [[1, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1]]
    assert torch.all(embeds[0, 6] == 0)


def test_tensor_event_path_skips_adaptor_and_pooling(tiny):
    ids = torch.tensor([[1, EVENT_TOKEN_INDEX, 5]])
    frames = torch.stack(_frames(1))
    emb = tiny.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                    event_tensors=frames)[4]
    assert emb.shape[1] == 2 + 5           # 2x2 patches + CLS for the 28px tower
    assert torch.allclose(emb[0, 1:6], tiny.visval_encode(frames)[0], atol=1e-5)


def test_generate_end_to_end_and_hidden_states(tiny):
    tok = ByteTokenizer(tiny.config.vocab_size)
    ids = tokenizer_event_token(prepare_event_prompt("Describe."), tok, return_tensors="pt")[None]
    frames = _frames(5)
    out = tiny.generate(ids, event_tensors=frames, event_image_sizes=[48, 64], do_sample=False,
                        max_new_tokens=6, eos_token_id=-1)
    assert out.shape == (1, 6)
    # batched tower == per-frame tower
    per = torch.stack([tiny.get_model().feature_adaptor(tiny.visval_encode(f[None]))[0] for f in frames])
    pooled = get_spatio_temporal_features(per)
    assert torch.allclose(pooled, tiny.encode_event_frames(frames), atol=1e-4)
    res = tiny.generate(ids, event_features=pooled[None], already_adapted=True, do_sample=False,
                        max_new_tokens=6, eos_token_id=-1, output_hidden_states=True,
                        return_dict_in_generate=True)
    assert torch.equal(res.sequences, out)
    n_prompt = ids.shape[1] - 1 + pooled.shape[0]
    assert len(res.hidden_states) == 6
    assert len(res.hidden_states[0]) == tiny.config.num_hidden_layers + 1
    assert res.hidden_states[0][-1].shape == (1, n_prompt, 64)
    assert res.hidden_states[3][-1].shape == (1, 1, 64)
    with pytest.raises(NotImplementedError):
        tiny.generate(ids)
    with pytest.raises(NotImplementedError):
        tiny.generate(ids, event_tensors=frames, inputs_embeds=torch.zeros(1))


def test_kv_cache_decode_equals_full_forward(tiny):
    torch.manual_seed(1)
    x = torch.randn(2, 9, 64)
    full = tiny.model(inputs_embeds=x).last_hidden_state
    part = tiny.model(inputs_embeds=x[:, :6], use_cache=True)
    assert part.past_key_values[0][0].shape == (2, 4, 6, 16)
    step = tiny.model(inputs_embeds=x[:, 6:], past_key_values=part.past_key_values, use_cache=True)
    assert torch.allclose(step.last_hidden_state, full[:, 6:], atol=1e-4)
    assert step.past_key_values[0][0].shape[2] == 9


def test_sampling_top_p_and_eos(tiny):
    tok = ByteTokenizer(tiny.config.vocab_size)
    ids = tokenizer_event_token(prepare_event_prompt("Q"), tok, return_tensors="pt")[None]
    frames = _frames(1)
    g = torch.Generator().manual_seed(0)
    a = tiny.generate(ids, event_tensors=frames, do_sample=True, temperature=0.8, top_p=0.9,
                      max_new_tokens=5, generator=g, eos_token_id=-1)
    g = torch.Generator().manual_seed(0)
    b = tiny.generate(ids, event_tensors=frames, do_sample=True, temperature=0.8, top_p=0.9,
                      max_new_tokens=5, generator=g, eos_token_id=-1)
    assert torch.equal(a, b)
    greedy = tiny.generate(ids, event_tensors=frames, do_sample=False, max_new_tokens=5, eos_token_id=-1)
    stop = tiny.generate(ids, event_tensors=frames, do_sample=False, max_new_tokens=5,
                         eos_token_id=int(greedy[0, 2]))
    assert stop.shape[1] == 3


def test_checkpoint_roundtrip_hf_layout(tiny, tmp_path):
    d = str(tmp_path / "ckpt")
    tiny.save_pretrained(d, max_shard_bytes=200_000)   # force several shards + index
    import json, os
    assert os.path.exists(os.path.join(d, "model.safetensors.index.json"))
    cfgj = json.load(open(os.path.join(d, "config.json")))
    assert cfgj["model_type"] == "EventChat_llama" and cfgj["event_feature_adaptor"] is True
    keys = set(json.load(open(os.path.join(d, "model.safetensors.index.json")))["weight_map"])
    for k in ("model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight",
              "model.layers.1.mlp.down_proj.weight", "model.norm.weight", "lm_head.weight",
              "model.visual_projector.0.weight", "model.visual_projector.2.bias",
              "model.feature_adaptor.weight",
              "model.visual_tower.visual_tower.vision_model.embeddings.class_embedding",
              "model.visual_tower.visual_tower.vision_model.pre_layrnorm.weight",
              "model.visual_tower.visual_tower.vision_model.encoder.layers.0.self_attn.out_proj.bias"):
        assert k in keys, k
    back = EventChatModel.from_pretrained(d, torch.float32)
    for (k, a), (_, b) in zip(tiny.state_dict().items(), back.state_dict().items()):
        assert torch.equal(a, b), k


def test_resize_token_embeddings_and_vision_tokenizer(tiny):
    import copy
    m = copy.deepcopy(tiny)
    tok = ByteTokenizer(m.config.vocab_size)

    class Args:
        mm_use_im_patch_token = True
        mm_use_im_start_end = True
        tune_mm_mlp_adapter = False
        pretrain_mm_mlp_adapter = None
    old = m.get_input_embeddings().weight.data.clone()
    m.initialize_vision_tokenizer(Args, tok)
    w = m.get_input_embeddings().weight.data
    assert w.shape[0] == old.shape[0] + 3 and m.lm_head.weight.shape[0] == w.shape[0]
    assert torch.equal(w[:old.shape[0]], old)
    assert torch.allclose(w[-1], w[:-2].mean(0), atol=1e-5)


def test_loss_with_labels(tiny):
    ids = torch.tensor([[1, EVENT_TOKEN_INDEX, 5, 6, 7]])
    labels = torch.tensor([[IGNORE_INDEX, IGNORE_INDEX, 5, 6, 7]])
    out = tiny(input_ids=ids, labels=labels, event_features=torch.randn(1, 3, 64), already_adapted=True)
    assert out.loss is not None and out.loss.item() > 0 and out.logits.shape[1] == 7


def test_sampler_top_k_then_top_p_cpu():
    """The reference op applies temperature, top-k (ties at the k-th value kept), then top-p."""
    from eventgpt_amd.ops import reference as ref
    g = torch.Generator().manual_seed(0)
    logits = torch.full((1, 50), -9.0)
    logits[0, [3, 10, 20, 30]] = torch.tensor([1.0, 0.5, 0.5, 0.0])
    seen = {int(ref.sample_top_p(logits, 1.0, 1.0, g, top_k=2)) for _ in range(200)}
    assert seen == {3, 10, 20}
    assert {int(ref.sample_top_p(logits, 1.0, 0.4, g, top_k=3)) for _ in range(50)} == {3}
    assert int(ref.sample_top_p(torch.randn(1, 99, generator=g), 0.7, 1.0, g, top_k=1).numel()) == 1
    wide = torch.randn(4, 300, generator=g)
    keep = wide.topk(5).indices
    for _ in range(40):
        pick = ref.sample_top_p(wide, 2.0, 1.0, g, top_k=5)
        assert all(int(pick[b]) in keep[b].tolist() for b in range(4))


def test_rope_qkv_split_and_column_blocks_on_cpu():
    """The packed q|k|v helpers the grouped attention path uses have plain-tensor fallbacks: the split +
    RoPE equals RoPE on the separate slices, and low-rank blocks added onto column ranges get exactly
    their slice of the gradient."""
    import torch
    from eventgpt_amd import ops
    torch.manual_seed(0)
    b, s, hq, hkv, d = 2, 7, 4, 2, 16
    qkv = torch.randn(b, s, (hq + 2 * hkv) * d, requires_grad=True)
    pos = torch.arange(s)[None].expand(b, s).contiguous()
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    ang = torch.arange(32).float()[:, None] * inv[None]
    cos, sin = ang.cos(), ang.sin()
    q, k, v = ops.rope_qkv_split(qkv, pos, cos, sin, hq, hkv, d)
    qs, ks, vs = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    assert torch.allclose(q, ops.apply_rope(qs.reshape(b, s, hq, d), pos, cos, sin))
    assert torch.allclose(k, ops.apply_rope(ks.reshape(b, s, hkv, d), pos, cos, sin))
    assert torch.equal(v, vs.reshape(b, s, hkv, d))
    (q.sum() + 2 * k.sum() + 3 * v.sum()).backward()
    assert qkv.grad is not None and torch.allclose(qkv.grad[..., -hkv * d:], torch.full((b, s, hkv * d), 3.0))

    y = torch.randn(b, s, 24, requires_grad=True)
    d0, d2 = torch.randn(b, s, 8, requires_grad=True), torch.randn(b, s, 4, requires_grad=True)
    base = y * 1.0                                      # a fresh tensor the helper may update in place
    out = ops.add_column_blocks(base, [d0, None, d2], [8, 12, 4])
    want = y.detach().clone()
    want[..., :8] += d0.detach()
    want[..., 20:] += d2.detach()
    assert torch.allclose(out, want)
    g = torch.randn_like(out)
    out.backward(g)
    assert torch.allclose(y.grad, g) and torch.allclose(d0.grad, g[..., :8]) and torch.allclose(d2.grad, g[..., 20:])
    with torch.no_grad():
        out2 = ops.add_column_blocks(y.detach().clone(), [d0, None, d2], [8, 12, 4])
    assert torch.allclose(out2, want)


def test_group_frozen_weights_is_a_gpu_feature():
    """On CPU tensors nothing is grouped and the attention / MLP blocks run their separate projections."""
    import torch
    from types import SimpleNamespace
    from eventgpt_amd.models.llama import LlamaMLP, group_frozen_weights
    mlp = LlamaMLP(SimpleNamespace(hidden_size=64, intermediate_size=128))
    for p in mlp.parameters():
        p.requires_grad_(False)
    assert group_frozen_weights((mlp.gate_proj, mlp.up_proj)) is None and not mlp.group_frozen()
    x = torch.randn(2, 5, 64)
    assert mlp(x).shape == (2, 5, 64)


def test_low_rank_update_on_cpu_and_through_the_lora_wrapper():
    """``ops.lowrank_delta`` has a plain-tensor fallback ((s x A^T) B^T with autograd's own gradients),
    and a wrapped layer's output is its base plus that update (dropout off)."""
    import torch
    from eventgpt_amd import ops
    from eventgpt_amd.align.lora import LoRALinear
    torch.manual_seed(0)
    x = torch.randn(2, 5, 32, requires_grad=True)
    a = torch.randn(8, 32, requires_grad=True)
    b = torch.randn(24, 8, requires_grad=True)
    y = ops.lowrank_delta(x, a, b, 0.5)
    ref = ((x @ a.t()) * 0.5) @ b.t()
    assert torch.allclose(y, ref, atol=1e-5)
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, (x, a, b), g)
    want = torch.autograd.grad(ref, (x, a, b), g)
    for u, v in zip(got, want):
        assert torch.allclose(u, v, atol=1e-5)
    lin = LoRALinear(torch.nn.Linear(32, 24), r=8, alpha=16, dropout=0.0)
    torch.nn.init.normal_(lin.lora_B.weight)
    xs = torch.randn(3, 32)
    want = lin.base(xs) + ((xs @ lin.lora_A.weight.t()) * 2.0) @ lin.lora_B.weight.t()
    assert torch.allclose(lin(xs), want, atol=1e-5)
    assert torch.allclose(lin.lora_delta(xs), want - lin.base(xs), atol=1e-5)
    lin.merge()
    assert lin.lora_delta(xs) is None and torch.allclose(lin(xs), want, atol=1e-4)
