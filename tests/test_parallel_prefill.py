"""Parallel-prefill benchmark: overlap model, acceptance, grid tiling, CLI on tiny CPU models."""
import json
import os

import numpy as np
import pytest

from eventgpt_amd.benchmarks import parallel_prefill as PP


def test_overlap_model_hand_case():
    e = {"prefill_complete_time": 0.10, "decode_tokens_per_sec": 100.0, "decode_tokens": 50, "stage5_time": 0.5,
         "total_time": 0.6}
    v = {"prefill_complete_time": 0.40, "decode_tokens": 40, "stage5_time": 0.8, "total_time": 1.2}
    m = PP.compute_parallel_metrics(e, v)
    assert m["overlap_window"] == pytest.approx(0.30)
    assert m["hidden_tokens"] == 30
    assert m["hidden_decode_time"] == pytest.approx(0.30)
    assert m["wall_clock_prefill"] == pytest.approx(0.40)
    assert m["wall_clock_decode"] == pytest.approx(0.8)          # max(0.5-0.3, 0.8)
    assert m["wall_clock_total"] == pytest.approx(1.2)
    assert m["parallel_speedup"] == pytest.approx(1.8 / 1.2)
    # drafter slower than verifier: no window, nothing hidden
    m2 = PP.compute_parallel_metrics(dict(e, prefill_complete_time=0.5), v)
    assert m2["overlap_window"] == 0 and m2["hidden_tokens"] == 0
    # window longer than the whole draft: capped by the tokens that exist
    m3 = PP.compute_parallel_metrics(dict(e, decode_tokens=10), dict(v, prefill_complete_time=5.0))
    assert m3["hidden_tokens"] == 10


def test_retokenized_acceptance():
    class Tok:
        def __call__(self, text):
            return {"input_ids": [len(w) for w in text.split()]}
    r = PP.retokenized_acceptance("a bb ccc dddd", "a bb xx dddd eeeee", Tok())
    assert r["method"] == "token_level_matching"
    # lengths [1,2,3,4] vs [1,2,2,4,5]: matches at 0,1,3 over a target of 5
    assert r["matched_tokens"] == 3 and r["acceptance_rate"] == pytest.approx(3 / 5) and r["draft_tokens"] == 4
    assert PP.retokenized_acceptance("", "x", Tok())["method"] == "token_level_mismatch"
    f = PP.retokenized_acceptance("The car turns left", "the car stops", None)
    assert f["method"] == "fallback_text_overlap" and f["acceptance_rate"] == pytest.approx(2 / 4)


def test_frames_to_grid():
    frames = [np.full((4, 6, 3), i, np.uint8) for i in range(5)]
    g = PP.frames_to_grid(frames, 2, 4)
    assert g.shape == (8, 24, 3)
    assert g[0, 0, 0] == 0 and g[0, 6, 0] == 1 and g[4, 0, 0] == 4 and g[4, 23, 0] == 4   # last frame repeats
    with pytest.raises(ValueError):
        PP.frames_to_grid([])


def test_cli_tiny_cpu(tmp_path):
    res = PP.main(["--random_init", "tiny", "--device", "cpu", "--max_samples", "2", "--warmup", "0",
                   "--max_new_tokens", "6", "--output_dir", str(tmp_path)])
    st = res["statistics"]
    assert st["n_samples"] == 2
    assert os.path.exists(res["paths"]["md"]) and os.path.exists(res["paths"]["json"])
    saved = json.load(open(res["paths"]["json"]))
    r0 = saved["results"][0]
    tt = r0["egpt"]["token_times"]
    assert len(tt["token_ids"]) == len(tt["token_times_ms"]) >= 1
    assert all(b >= a for a, b in zip(tt["token_times_ms"], tt["token_times_ms"][1:]))
    # the timed decode and the stage run are the same greedy decode
    assert tt["token_ids"] == r0["egpt"]["token_ids"][: len(tt["token_ids"])]
    assert r0["parallel"]["wall_clock_total"] <= r0["parallel"]["sequential_total"] + 1e-9
    assert "live" not in r0                                       # CPU: modelled overlap only
    assert "Overlap (modelled" in open(res["paths"]["md"]).read()
