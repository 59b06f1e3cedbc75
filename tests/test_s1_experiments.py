"""Shared-decoder injection, token-pair extraction and alpha sweeps on tiny models (CPU)."""
import json

import pytest
import torch

from eventgpt_amd.benchmarks import s1_experiments as S1
from eventgpt_amd.data.tokenizer import ByteTokenizer
from eventgpt_amd.models import EventChatConfig, EventChatModel
from eventgpt_amd.models.video_llava import VideoLlavaConfig, VideoLlavaModel, VideoProcessor
from eventgpt_amd.pipeline import dataset as D
from eventgpt_amd.pipeline.extract_hidden_states import HiddenStateExtractor


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    data = D.make_synthetic_dataset(str(tmp_path_factory.mktemp("ds")), n_samples=3, n_events=3000)
    egpt = EventChatModel.from_config(EventChatConfig.tiny(), torch.float32, "cpu", seed=0)
    vcfg = VideoLlavaConfig.tiny()
    vl = VideoLlavaModel.from_config(vcfg, torch.float32, "cpu", seed=1)
    tok = ByteTokenizer()
    return D.InstructionDataset(data), egpt, vl, VideoProcessor(vcfg, tok), tok


def test_shared_decoder_pipeline(setup):
    ds, egpt, vl, proc, _ = setup
    pipe = S1.SharedDecoderPipeline(egpt, vl, proc)
    f = pipe.extract_eventgpt_features(ds.event_image_paths(ds[0]))
    assert f.shape == (vl.config.tokens_per_frame, 64)
    assert S1._resample_tokens(torch.arange(4.0)[:, None], 7).shape == (7, 1)
    assert torch.allclose(S1._resample_tokens(torch.ones(5, 3), 9), torch.ones(9, 3))
    res = pipe.run(ds, "What?", max_samples=3, max_new_tokens=6, num_epochs=30)
    assert res["n"] == 3 and res["adapter_losses"][-1] < res["adapter_losses"][0]
    assert res["cos_after"] > res["cos_before"]
    assert 0 <= res["accept_adapted_mean"] <= 1 and len(res["samples"]) == 3
    # injecting the verifier's own (mean-pooled) features twice gives identical text
    from eventgpt_amd.models.video_llava import load_video_frames
    v = pipe.extract_videollava_features(load_video_frames(ds.video_source(ds[0]), vl.config.num_frames))
    a = pipe.generate_with_shared_decoder(v, "What?", 5)
    b = pipe.generate_with_shared_decoder(v, "What?", 5)
    assert a[1] == b[1] and len(a[1]) == 5 and a[2] > 0
    ad = S1.FeatureAlignmentAdapter(S1.AdapterConfig(8, 8, 16, 2, 0.0))
    assert ad.use_residual and not S1.FeatureAlignmentAdapter(S1.AdapterConfig(8, 6, 16, 2)).use_residual


def test_token_pairs_and_alpha(setup, tmp_path):
    ds, egpt, vl, proc, tok = setup
    ex = HiddenStateExtractor(egpt, tok, vl, proc, max_new_tokens=6)
    out = S1.extract_token_pairs(ex, ds, ["What?", "Why?"], str(tmp_path / "tok.json"), max_samples=2,
                                 with_hidden=True)
    assert out["num_pairs"] == 4 and len(out["hidden"]["egpt"]) == 4 and out["hidden"]["vl"][0].shape == (6, 64)
    again = S1.extract_token_pairs(ex, ds, ["What?", "Why?"], str(tmp_path / "tok.json"), max_samples=3)
    assert again["num_pairs"] == 6                                   # resumed: only the third scene was new
    from eventgpt_amd.align.token_align import TokenAlignmentDataset
    tds = TokenAlignmentDataset.from_json(str(tmp_path / "tok.json"), max_seq_len=8)
    assert len(tds) == 6
    pairs = [{"egpt_tokens": [1, 2, 3, 4, 5, 6], "vl_tokens": [1, 2, 9, 4, 5, 6]}]
    r = S1.benchmark_alpha(pairs, gammas=(2, 5), c_ratio=0.1)
    assert r["alpha"] == pytest.approx(5 / 6)
    # gamma 5: window at 0 accepts 2, target fixes pos 2, window at 3 accepts 3 -> 2 steps, 2.5 accepted per step
    assert r["per_gamma"]["5"]["mean_accepted_per_step"] == 2.5 and r["per_gamma"]["5"]["tokens_per_target_call"] == 3.0
    # gamma 2: (0: 2) -> 3 (3: 2) -> 6  => 2 steps of 2
    assert r["per_gamma"]["2"]["mean_accepted_per_step"] == 2.0
    assert r["per_gamma"]["2"]["theoretical_speedup"] == pytest.approx(S1.calculate_speculative_speedup(0.1, 5 / 6, 2))
    assert S1.consecutive_prefix([1, 2, 3], [1, 2, 4], 0, 5) == 2
