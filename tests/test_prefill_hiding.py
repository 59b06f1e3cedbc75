"""Prefill-hiding runtimes: analytic helpers (CPU) and live runs (GPU)."""
import pytest
import torch

from eventgpt_amd.runtime import prefill_hiding as P


def test_analytic_helpers():
    b = P.compute_prefill_benefit(100.0, 300.0, 20.0)
    assert b["overlap_window_ms"] == 200.0 and b["free_tokens"] == 10 and b["prefill_speedup"] == 3.0
    assert P.compute_prefill_benefit(300.0, 100.0)["free_tokens"] == 0
    t = P.compute_theoretical_speedup(100.0, 300.0, vl_token_time_ms=10.0, max_new_tokens=50,
                                      acceptance_rate=0.5, egpt_token_time_ms=10.0, verify_time_ms=20.0)
    assert t["free_drafts"] == 20 and t["accepted"] == 10 and t["vl_baseline_ms"] == 800.0
    assert t["sd_time_ms"] == 300 + 20 + 400 and t["e2e_speedup"] == pytest.approx(800 / 720)
    ev = P.evaluate_sd_outputs(
        [{"output_tokens": [1, 2, 3], "wall_clock_time": 0.05, "n_drafted": 4, "n_accepted": 2}],
        [{"tokens": [1, 2, 3, 4], "total_ms": 100.0}])
    assert ev["lossless"] and ev["speedup"] == pytest.approx(2.0) and ev["accept_rate"] == 0.5


@pytest.mark.gpu
def test_live_runtimes(dev):
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.align.token_align import TokenAdapterForSD
    from eventgpt_amd.data.tokenizer import ByteTokenizer
    from eventgpt_amd.models import EventChatConfig, EventChatModel

    def cfg(layers):
        return EventChatConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=layers,
                               num_attention_heads=4, max_position_embeddings=1024)
    target = EventChatModel.from_config(cfg(3), torch.bfloat16, dev, seed=21)
    drafter = EventChatModel.from_config(cfg(1), torch.bfloat16, dev, seed=22)
    d, v = drafter.engine(max_len=1024, max_new_tokens=128), target.engine(max_len=1024, max_new_tokens=128)
    g = torch.Generator(device=dev).manual_seed(0)
    d_emb = (torch.randn(24, 512, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    v_emb = (torch.randn(400, 512, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    ref = v.generate(v_emb, 20, eos_token_id=-1).tolist()
    tok = ByteTokenizer(512)

    po = P.PrefillOnlyInference(d, v, tok, tok, max_new_tokens=20).generate(d_emb, v_emb, eos_token_id=-1)
    assert po.output_tokens == ref and len(po.preview_tokens) == 20
    assert po.wall_clock_time > 0 and po.total_speedup_vs_sequential > 0.5 and po.vl_prefill_time > 0

    table = torch.arange(512, device=dev)
    ptv = P.PrefillThenVerifyInference(d, v, TokenAdapterForSD(None, table), tok, 20, 8)
    r = ptv.generate(d_emb, v_emb, eos_token_id=-1)
    assert r.output_tokens == ref and 1 <= len(r.draft_tokens) <= 8 and r.n_drafted == len(r.draft_tokens) - 1
    # the verifier drafting for itself through the same interface accepts every free token
    twin = EventChatModel.from_config(cfg(3), torch.bfloat16, dev, seed=21)
    r2 = P.PrefillThenVerifyInference(twin.engine(max_len=1024, max_new_tokens=128), v, None, tok, 20, 6) \
        .generate(v_emb, v_emb, eos_token_id=-1)
    assert r2.output_tokens == ref and r2.n_accepted == r2.n_drafted

    torch.manual_seed(0)
    ad = create_adapter(1, 512, bottleneck_dim=32)
    r3 = P.AlignedPrefillInference(d, v, ad, target.lm_head.weight.data, verifier_tokenizer=tok,
                                   max_new_tokens=20, max_draft_tokens=8).generate(d_emb, v_emb, eos_token_id=-1)
    assert r3.output_tokens == ref
    ev = P.evaluate_sd_outputs([r.as_dict(), r3.as_dict()], [{"tokens": ref, "total_ms": 50.0}] * 2)
    assert ev["lossless"] and ev["n"] == 2


def test_prefill_hiding_runner_theory_mode(tmp_path):
    from eventgpt_amd.benchmarks import prefill_hiding_bench as PH
    res = PH.main(["--theory", "--egpt_prefill_ms", "100", "--vl_prefill_ms", "300", "--vl_token_ms", "20",
                   "--egpt_token_ms", "10", "--verify_ms", "50", "--acceptance_rate", "0.5",
                   "--max_new_tokens", "50", "--output_dir", str(tmp_path)])
    row = next(t for t in res["theory"] if t["acceptance_rate"] == 0.5)
    # window 200 ms -> 20 free drafts, 10 accepted, 200 ms saved of 1300; verify costs 50
    assert row["free_drafts"] == 20 and row["accepted"] == 10 and row["time_saved_ms"] == 200
    assert row["vl_baseline_ms"] == 1300 and row["sd_time_ms"] == 1150
    assert abs(row["e2e_speedup"] - 1300 / 1150) < 1e-9
    md = open(res["paths"]["md"]).read()
    assert "Closed-form estimate" in md and "0.500" in md
    agg = PH.aggregate([])
    assert agg == {"n": 0}


@pytest.mark.gpu
def test_prefill_hiding_runner_gpu(tmp_path):
    from eventgpt_amd.benchmarks import prefill_hiding_bench as PH
    res = PH.main(["--random_init", "small", "--max_samples", "2", "--warmup", "1", "--max_new_tokens", "12",
                   "--max_draft_tokens", "8", "--output_dir", str(tmp_path)])
    agg = res["aggregate"]
    assert agg["n"] == 2
    for m in ("prefill_only", "prefill_then_verify", "aligned"):
        assert m in agg, m
        # every runtime returns the verifier's own greedy tokens
        assert agg[m]["identical_outputs"] == 2, (m, agg[m])
    assert agg["prefill_then_verify"]["lossless"] and "theory" in res
