"""Multimodal splice on the GPU: the one-launch gather kernel (inference) against the
differentiable tensor path (training) and a per-sample construction of the same layout."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_splice_rows_kernel(dev):
    from eventgpt_amd.ops import hip
    g = torch.Generator(device=dev).manual_seed(1)
    embed = torch.randn(300, 512, device=dev, generator=g).to(torch.bfloat16)
    feats = torch.randn(40, 512, device=dev, generator=g).to(torch.bfloat16)
    pad = -(1 << 30)
    m = torch.tensor([[5, -1, -2, -40, 299, pad, pad], [0, 1, 2, 3, -7, -8, 7]], dtype=torch.int32, device=dev)
    out = hip.raw.splice_rows(embed, feats, m)
    assert out.shape == (2, 7, 512)
    for b in range(2):
        for l in range(7):
            v = int(m[b, l])
            want = embed[v] if v >= 0 else (feats[-1 - v] if v > pad else torch.zeros_like(embed[0]))
            assert torch.equal(out[b, l], want), (b, l)


def test_prepare_inputs_kernel_path_equals_tensor_path(dev):
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.models.event_chat import EVENT_TOKEN_INDEX, IGNORE_INDEX
    cfg = EventChatConfig(vocab_size=600, hidden_size=512, intermediate_size=1408, num_hidden_layers=1,
                          num_attention_heads=4, max_position_embeddings=512)
    model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    g = torch.Generator(device=dev).manual_seed(2)
    ids = torch.randint(3, 600, (3, 12), device=dev, generator=g)
    ids[0, 4] = EVENT_TOKEN_INDEX
    ids[1, 0] = EVENT_TOKEN_INDEX
    ids[2, 11] = EVENT_TOKEN_INDEX
    mask = torch.ones_like(ids)
    mask[1, 9:] = 0
    labels = ids.clone()
    labels[:, :3] = IGNORE_INDEX
    feats = [torch.randn(n, 512, device=dev, generator=g).to(torch.bfloat16) for n in (9, 5, 7)]
    pos = torch.arange(12, device=dev)[None].expand(3, 12)
    call = lambda: model.prepare_inputs_labels_for_multimodal(ids, pos, mask, None, labels, event_features=feats,  # noqa: E731
                                                              already_adapted=True)
    with torch.no_grad():
        a = call()                                   # gather kernel
    model.get_model().embed_tokens.weight.requires_grad_(True)
    with torch.enable_grad():
        b = call()                                   # differentiable tensor expressions
    assert b[4].requires_grad and not a[4].requires_grad
    for x, y in zip(a, b):
        if x is None:
            assert y is None
            continue
        assert x.shape == y.shape and torch.equal(x.float(), y.detach().float())
    embeds, out_mask, out_labels = a[4], a[2], a[5]
    # sample 1: 5 event rows first, then the 8 kept prompt tokens (mask drops the last 3)
    assert int(out_mask[1].sum()) == 5 + 8
    assert torch.equal(embeds[1, :5], feats[1])
    assert torch.equal(embeds[1, 5:13], model.get_model().embed_tokens.weight.detach()[ids[1, 1:9]])
    assert float(embeds[1, 13:].abs().sum()) == 0
    assert out_labels[1, :5].eq(IGNORE_INDEX).all() and torch.equal(out_labels[1, 7:13], labels[1, 3:9])


def test_splice_with_trainable_features_is_one_node(dev):
    """Frozen token table + event features that train (projector / LoRA tuning): the splice is the gather
    kernel with a row-gather backward; embeddings and feature gradients equal the tensor expressions'."""
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.models.event_chat import EVENT_TOKEN_INDEX
    cfg = EventChatConfig(vocab_size=600, hidden_size=512, intermediate_size=1408, num_hidden_layers=1,
                          num_attention_heads=4, max_position_embeddings=512)
    model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=0)
    emb_w = model.get_model().embed_tokens.weight
    g = torch.Generator(device=dev).manual_seed(2)
    ids = torch.randint(3, 600, (3, 12), device=dev, generator=g)
    ids[0, 4] = EVENT_TOKEN_INDEX
    ids[1, 0] = EVENT_TOKEN_INDEX
    ids[2, 11] = EVENT_TOKEN_INDEX
    mask = torch.ones_like(ids)
    mask[1, 9:] = 0
    pos = torch.arange(12, device=dev)[None].expand(3, 12)

    def run(table_trains):
        emb_w.requires_grad_(table_trains)
        emb_w.grad = None
        feats = [torch.randn(n, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(5 + n))
                 .to(torch.bfloat16).requires_grad_(True) for n in (9, 5, 7)]
        out = model.prepare_inputs_labels_for_multimodal(ids, pos, mask, None, None, event_features=feats,
                                                         already_adapted=True)[4]
        go = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(9)).to(out.dtype)
        out.backward(go)
        return out, [f.grad for f in feats], type(out.grad_fn).__name__

    e1, g1, n1 = run(False)
    e2, g2, n2 = run(True)
    assert n1 == "_SpliceRowsBackward" and n2 != n1
    assert torch.equal(e1, e2.detach())
    for a, b in zip(g1, g2):
        assert a is not None and torch.equal(a, b)
    emb_w.requires_grad_(False)
