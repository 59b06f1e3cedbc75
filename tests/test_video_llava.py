"""Video verifier: HF-layout state dict, feature-layer selection, placeholder merge, generate."""
import numpy as np
import pytest
import torch

from eventgpt_amd.data.tokenizer import ByteTokenizer
from eventgpt_amd.models.video_llava import (VideoLlavaConfig, VideoLlavaModel, VideoProcessor, load_video_frames,
                                             sample_frame_indices)


@pytest.fixture(scope="module")
def model():
    return VideoLlavaModel.from_config(VideoLlavaConfig.tiny(), torch.float32, "cpu", seed=2)


def test_state_dict_layout_and_roundtrip(model, tmp_path):
    keys = set(model.state_dict())
    for k in ("video_tower.vision_model.embeddings.patch_embedding.weight",
              "video_tower.vision_model.pre_layrnorm.weight",
              "multi_modal_projector.linear_1.weight", "multi_modal_projector.linear_2.bias",
              "language_model.model.embed_tokens.weight",
              "language_model.model.layers.0.self_attn.q_proj.weight", "language_model.lm_head.weight"):
        assert k in keys, k
    model.save_pretrained(str(tmp_path / "vl"))
    again = VideoLlavaModel.from_pretrained(str(tmp_path / "vl"), torch.float32)
    assert not again.load_report["missing"] and not again.load_report["unexpected"]
    for (k, a), b in zip(model.state_dict().items(), again.state_dict().values()):
        assert torch.equal(a, b), k
    assert again.config.num_frames == 2 and again.config.vision_feature_layer == -2


def test_features_use_second_to_last_layer_and_all_tokens(model):
    cfg = model.config
    px = torch.randn(1, cfg.num_frames, 3, 28, 28)
    f = model.get_video_features(px)
    assert f.shape == (1, cfg.num_frames * cfg.tokens_per_frame, cfg.text_config.hidden_size)
    hs = model.video_tower(px[0], output_hidden_states=True).hidden_states
    assert len(hs) == cfg.vision_config.num_hidden_layers + 1
    ref = model.multi_modal_projector(hs[-2]).reshape(1, -1, f.shape[-1])
    assert torch.allclose(f, ref, atol=1e-6)
    assert not torch.allclose(f, model.multi_modal_projector(hs[-1]).reshape(1, -1, f.shape[-1]), atol=1e-5)


def test_merge_generate_and_processor(model):
    cfg = model.config
    tok = ByteTokenizer()
    proc = VideoProcessor(cfg, tok)
    frames = [np.random.randint(0, 255, (40, 60, 3), dtype=np.uint8) for _ in range(cfg.num_frames)]
    inputs = proc("hi", frames)
    ids = inputs["input_ids"]
    assert int((ids == cfg.video_token_index).sum()) == 1
    assert inputs["pixel_values_videos"].shape == (1, cfg.num_frames, 3, 28, 28)
    emb = model.prepare_embeds(ids, inputs["pixel_values_videos"])
    assert emb.shape[1] == ids.shape[1] - 1 + cfg.video_tokens
    # pre-expanded placeholders give the same embeddings
    at = int(torch.nonzero(ids[0] == cfg.video_token_index)[0])
    wide = torch.cat([ids[0, :at], torch.full((cfg.video_tokens,), cfg.video_token_index), ids[0, at + 1:]])[None]
    assert torch.equal(model.prepare_embeds(wide, inputs["pixel_values_videos"]), emb)
    out = model.generate(ids, inputs["pixel_values_videos"], max_new_tokens=5, eos_token_id=-1,
                         output_hidden_states=True, return_dict_in_generate=True)
    assert out.sequences.shape == (1, 5) and len(out.hidden_states) == 5
    # teacher-forced forward reproduces the greedy choices
    full = torch.cat([emb, model.get_input_embeddings()(out.sequences[:, :-1])], 1)
    logits = model(inputs_embeds=full).logits
    assert torch.equal(logits[0, emb.shape[1] - 1:].argmax(-1), out.sequences[0])


def test_frame_sampling(tmp_path):
    assert sample_frame_indices(100, 8) == np.linspace(0, 99, 8, dtype=int).tolist()
    assert sample_frame_indices(3, 5) == [0, 1, 2, 2, 2] and sample_frame_indices(0, 4) == []
    np.save(tmp_path / "clip.npy", np.zeros((10, 8, 8, 3), dtype=np.uint8))
    assert len(load_video_frames(str(tmp_path / "clip.npy"), 4)) == 4
