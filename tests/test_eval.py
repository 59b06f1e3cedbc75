"""Acceptance metrics, two-phase evaluation, profilers/timers, beam search (CPU)."""
import json
import math
import os

import pytest
import torch

from eventgpt_amd.eval import acceptance as A
from eventgpt_amd.eval import profiler as P
from eventgpt_amd.eval import reports, timers, two_phase


def _unit(rows):
    return torch.nn.functional.normalize(torch.tensor(rows, dtype=torch.float32), dim=-1)


def _pair_with_cos(cos):
    """[N,S,2] pairs whose cosine is exactly the given [N,S] matrix."""
    c = torch.tensor(cos, dtype=torch.float32)
    a = torch.stack([torch.ones_like(c), torch.zeros_like(c)], -1)
    b = torch.stack([c, (1 - c * c).clamp_min(0).sqrt()], -1)
    return a, b


def test_feature_metrics_hand_case():
    cos = [[0.99, 0.96, 0.70, 0.99], [0.93, 0.50, 0.99, 0.99], [0.99, 0.99, 0.99, 0.99]]
    a, b = _pair_with_cos(cos)
    lens = torch.tensor([4, 3, 2])
    mask = A.seq_mask(lens, 4)
    t = A.TimingConfig()
    m, cs, valid = A.compute_all_metrics_parallel(a, b, mask, t, A.SDConfig(gamma_decode=3))
    assert torch.allclose(cs, torch.tensor(cos), atol=1e-5)
    assert m["total_valid_positions"] == 9 and valid.numel() == 9
    # >0.90 among valid: row0 3, row1 2 (0.93, 0.99), row2 2  -> 7/9
    assert m["accept_90"] == pytest.approx(7 / 9)
    assert m["accept_95"] == pytest.approx(6 / 9)
    # runs @0.90: 2, 1, min(4, len 2) = 2
    assert m["consecutive_mean_90"] == pytest.approx(5 / 3)
    assert m["consecutive_max_90"] == 2 and m["consecutive_min_90"] == 1
    assert m["num_hidden_tokens_mean"] == pytest.approx(3.0)
    assert m["sd_accept_rate_prefill_gamma"] == pytest.approx(5 / 9)
    # first 3 positions, masked: rows give 2, 1, 2
    assert m["sd_consecutive_gamma5"] == pytest.approx(5 / 3)
    assert m["sd_accept_rate_decode_gamma5"] == pytest.approx(5 / 9)
    free = (310 - 130) / 25
    assert m["prefill_free_draft_tokens"] == pytest.approx(free)
    assert m["prefill_effective_tokens"] == pytest.approx(5 / 3)
    assert m["speedup_prefill"] == pytest.approx(1 + (5 / 3) * 25 / 310)
    assert m["decode_overhead_ratio"] == pytest.approx(0.06)
    assert m["speedup_decode_gamma5"] == pytest.approx((5 / 3 + 1) / 1.06)
    base = 310 + 50 * 25
    sd = 310 + 50 / (5 / 3 + 1) * (25 * 4 + 1.5)
    assert m["e2e_baseline_time_ms"] == base and m["e2e_sd_time_ms"] == pytest.approx(sd)
    assert m["speedup_e2e"] == pytest.approx(base / sd)
    assert m["position_accept_90"] == pytest.approx([1.0, 2 / 3, 0.5, 1.0])
    stats = A.compute_per_position_stats(cs, mask)
    assert [s["num_samples"] for s in stats] == [3, 3, 2, 1]
    sb = A.stage_breakdown(m, t)
    assert sb["sd_total_ms"] == pytest.approx(sd) and sb["baseline_total_ms"] == base


def test_token_metrics_and_projection():
    torch.manual_seed(0)
    V, H = 50, 16
    W = torch.randn(V, H)
    hid = torch.randn(5, 6, H)
    tok, topk = A.project_to_tokens(hid, W, None, top_k=10, batch_size=2, device="cpu")
    ref = hid @ W.T
    assert torch.equal(tok, ref.argmax(-1)) and torch.equal(topk, ref.topk(10, -1).indices)
    tok1, _ = A.project_to_tokens(hid, W, None, top_k=1, batch_size=3, device="cpu")
    assert torch.equal(tok1, tok)

    vl = torch.tensor([[1, 2, 3, 4], [5, 6, 7, 8]])
    al = torch.tensor([[1, 2, 9, 4], [0, 6, 7, 8]])
    al_topk = al.unsqueeze(-1).repeat(1, 1, 10)
    al_topk[0, 2, 3] = 3       # the miss at (0,2) is a top-5 hit
    al_topk[1, 0, 7] = 5       # the miss at (1,0) is only a top-10 hit
    mask = A.seq_mask(torch.tensor([4, 3]), 4)
    m = A.token_metrics_from_ids(al, al_topk, vl, mask)
    assert m["token_match_rate"] == pytest.approx(5 / 7)
    assert m["token_match_rate_top5"] == pytest.approx(6 / 7)
    assert m["token_match_rate_top10"] == pytest.approx(7 / 7)
    assert m["token_consecutive_mean"] == pytest.approx(1.0)      # runs 2 and 0
    assert m["position_token_accept"] == pytest.approx([0.5, 1.0, 0.5, 1.0])

    t = A.TimingConfig()
    tp = A.compute_two_phase_sd_metrics(al, vl, mask, t, gamma_decode=3)
    assert tp["token_sd_gamma_prefill"] == 7
    assert tp["token_sd_prefill_consecutive_mean"] == pytest.approx(1.0)
    assert tp["token_sd_prefill_accept_rate"] == pytest.approx(0.25)   # window clipped to S=4
    assert tp["token_sd_decode_consecutive_mean"] == pytest.approx(1.0)
    sd = 310 + 50 / 2 * (25 * 4 + 1.5)
    assert tp["token_sd_e2e_sd_ms"] == pytest.approx(sd)

    cos = torch.tensor([[0.99, 0.99, 0.5, 0.99], [0.5, 0.97, 0.97, 0.97]])
    c = A.compute_cosim_vs_token_correlation(cos, al == vl, mask)
    assert c["cosim_90_count"] == 5 and c["cosim_90_token_match"] == 1.0
    assert c["cosim_token_pearson_r"] == pytest.approx(1.0, abs=1e-5) or c["cosim_token_pearson_r"] > 0.9


def test_two_phase_functions():
    vl = torch.tensor([[1, 2, 3, 4, 5, 6], [7, 8, 9, 10, 11, 12]])
    mask = torch.ones(2, 6)
    t = A.TimingConfig()
    pre = torch.tensor([[1, 2, 3, 0, 5, 6], [7, 0, 9, 10, 11, 12]])
    p1 = two_phase.phase1_prefill(pre, vl, mask, 7, t)
    assert p1["phase1_prefill_top1"] == pytest.approx(10 / 12)
    assert p1["phase1_free_tokens"] == pytest.approx(2.0) and p1["phase1_time_saved_ms"] == 50.0
    # decode adapter predicts the NEXT token
    dec = torch.tensor([[2, 3, 4, 0, 6, 0], [8, 9, 0, 11, 12, 0]])
    topk = dec.unsqueeze(-1).repeat(1, 1, 10)
    topk[0, 3, 4] = 5
    p2 = two_phase.phase2_decode(dec, topk, vl, mask, gamma_decode=4)
    assert p2["phase2_same_pos_top1"] == 0.0
    assert p2["phase2_shifted_top1"] == pytest.approx(8 / 10)
    assert p2["phase2_shifted_top5"] == pytest.approx(9 / 10)
    assert p2["phase2_decode_consecutive"] == pytest.approx(2.5)      # runs 3 and 2 in the first 4
    sp = two_phase.combined_speedup(2.0, 2.5, t)
    sd_ms = 310 + (48 / 3.5) * 26.5
    assert sp["sd_time_ms"] == pytest.approx(sd_ms) and sp["e2e_speedup"] == pytest.approx(1560 / sd_ms)
    assert two_phase.combined_speedup(0.0, 0.0, t)["e2e_speedup"] == pytest.approx(1.0)


def test_cli_end_to_end(tmp_path):
    from eventgpt_amd.align.adapters import create_adapter
    from eventgpt_amd.align.store import ChunkedHiddenStateWriter
    torch.manual_seed(1)
    H, V = 32, 40
    w = ChunkedHiddenStateWriter(str(tmp_path / "data"), chunk_size=4)
    for i in range(6):
        n = 5 + i % 3
        e = torch.randn(n, H)
        w.add_sample(e, e + 0.05 * torch.randn(n, H), {"sample_idx": i, "question_idx": 0})
    w.flush()
    torch.save({"lm_head_weight": torch.randn(V, H)}, tmp_path / "lm.pt")
    ck1, ck2 = str(tmp_path / "l1.pt"), str(tmp_path / "l5f.pt")
    create_adapter(1, H, bottleneck_dim=8).save_checkpoint(ck1)
    create_adapter(6, H, num_heads=2, ffn_dim=16).save_checkpoint(ck2)
    out = tmp_path / "out"
    m = A.main(["--checkpoint", ck1, "--test_data", str(tmp_path / "data"), "--lm_head", str(tmp_path / "lm.pt"),
                "--output_dir", str(out), "--device", "cpu", "--batch_size", "4"])
    for f in ("acceptance_metrics.json", "position_stats.json", "metrics_comparison.md", "metrics_summary.png",
              "stage_timeline.png", "token_metrics.png"):
        assert (out / f).exists(), f
    saved = json.load(open(out / "acceptance_metrics.json"))
    assert saved["batch_size"] == 6 and "token_match_rate" in saved and "speedup_e2e" in saved
    assert 0.0 <= m["cos_sim_mean"] <= 1.0
    r = two_phase.main(["--prefill_checkpoint", ck1, "--decode_checkpoint", ck2, "--test_data",
                        str(tmp_path / "data"), "--lm_head", str(tmp_path / "lm.pt"), "--label", "t",
                        "--output_dir", str(out), "--device", "cpu"])
    assert (out / "two_phase_results_t.json").exists()
    assert r["decode_adapter_class"] == "FusedEAGLEAdapter" and r["gamma_prefill"] == 7
    r2 = two_phase.main(["--decode_checkpoint", ck1, "--no_prefill", "--decode_vlm_only", "--test_data",
                         str(tmp_path / "data"), "--lm_head", str(tmp_path / "lm.pt"),
                         "--output_dir", str(out), "--device", "cpu"])
    assert (out / "decode_only_results.json").exists() and r2["phase1_free_tokens"] == 0.0


def test_profilers(capsys):
    import time
    P.set_color_enabled(False)
    assert not P.is_color_enabled() and P.colorize_text("x", P.Colors.RED) == "x"
    with P.Profiler("sleep") as p:
        time.sleep(0.01)
    assert 0.009 < p.get_duration() < 0.5 and p.get_duration_ms() == pytest.approx(p.get_duration() * 1e3)
    calls = []
    ap = P.AveragingProfiler("f", num_runs=3, warmup=1)
    assert ap.profile_function(lambda x: calls.append(x) or x * 2, 4) == 8
    assert len(calls) == 4 and len(ap.get_all_durations()) == 3 and ap.get_average_duration_ms() >= 0

    @P.profile_function("deco")
    def f(a, b=1):
        return a + b

    @P.profile_function_avg(num_runs=2)
    def g(a):
        return a * 3
    assert f(1, b=2) == 3 and g(2) == 6
    with P.time_block("blk"):
        pass
    assert P.profile_step("s", lambda a: a + 1, 1) == 2
    assert P.profile_step_avg("s", lambda a: a + 1, 2, 1) == 2
    ms = P.MultiStepProfiler("proc")
    ms.start_process()
    with ms.step("a"):
        time.sleep(0.005)
    with ms.step("b"):
        pass
    ms.add_step_result("c", 0.5)
    res = ms.end_process()
    assert set(res) == {"a", "b", "c"} and res["a"] >= 0.004 and ms.total >= res["a"]
    P.set_color_enabled(True)
    assert P.colorize_duration(2.0).startswith(P.Colors.BRIGHT_YELLOW)
    assert "[PROFILING]" in capsys.readouterr().out


def test_stage_timer_and_hooks():
    import time
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    st = timers.StageTimer("cpu")
    with st.stage("load"):
        time.sleep(0.004)
    with st.stage("load"):
        time.sleep(0.004)
    st.add("x", 1.5)
    r = st.result()
    assert r["load"] >= 7.0 and r["x"] == 1.5
    model = EventChatModel.from_config(EventChatConfig.tiny(), torch.float32, "cpu", seed=0)
    vh = timers.VisionTimingHooks(model)
    assert vh.register()
    size = model.config.vision_config.image_size
    model.visval_encode(torch.randn(1, 3, size, size))
    assert vh.vision_time > 0
    vh.unregister()
    ph = timers.PrefillDecodeHooks(model)
    assert ph.register_hooks()
    emb = torch.randn(1, 6, model.config.hidden_size)
    model.generate_from_embeds(emb, max_new_tokens=4, eos_token_id=-1, use_engine=False)
    ph.finalize_decode_time()
    assert ph.forwards == 4 and ph.prefill_time > 0 and ph.decode_time > 0
    ph.unregister_hooks()
    out = model.model(inputs_embeds=emb, use_cache=True)
    mb = timers.estimate_kv_cache_mb(out.past_key_values)
    cfg = model.config
    expect = 2 * cfg.num_hidden_layers * 6 * cfg.num_key_value_heads * (cfg.hidden_size // cfg.num_attention_heads) * 4
    assert mb == pytest.approx(expect / 2 ** 20)
    assert timers.model_memory_mb(model) > 0 and timers.get_gpu_memory_mb() >= 0


def test_beam_search_matches_exhaustive():
    from eventgpt_amd.models import EventChatConfig, EventChatModel
    from eventgpt_amd.runtime.beam import beam_search
    cfg = EventChatConfig.tiny()
    model = EventChatModel.from_config(cfg, torch.float32, "cpu", seed=3)
    emb = torch.randn(1, 5, cfg.hidden_size)
    greedy = model.generate_from_embeds(emb, max_new_tokens=3, eos_token_id=-1, use_engine=False)
    one = beam_search(model, emb, None, 1, 3, -1)
    assert torch.equal(one, greedy)
    # with K = V the beam is exhaustive over 2-token continuations: compare with brute force
    V = cfg.vocab_size
    lp1 = torch.log_softmax(model(inputs_embeds=emb).logits[0, -1].float(), -1)
    best, best_seq = -1e9, None
    top = lp1.topk(8).indices.tolist()
    for a in top:
        e2 = torch.cat([emb, model.model.embed_tokens(torch.tensor([[a]]))], 1)
        lp2 = torch.log_softmax(model(inputs_embeds=e2).logits[0, -1].float(), -1)
        s, b = lp2.max(0)
        if float(lp1[a] + s) > best:
            best, best_seq = float(lp1[a] + s), [a, int(b)]
    got = model.generate_from_embeds(emb, num_beams=8, max_new_tokens=2, eos_token_id=-1)
    assert got[0].tolist() == best_seq
