"""E2E wall-clock benchmark: config building / discovery / aggregation on CPU, full run on GPU."""
import json
import os

import pytest
import torch

from eventgpt_amd.benchmarks import e2e_wallclock as E


def test_find_adapter_checkpoints_layouts(tmp_path):
    for rel in ("L1/L1_20260101_000000", "L1/L1_20260202_000000", "L5F/L5F_20260101_000000"):
        os.makedirs(tmp_path / "a" / rel)
        (tmp_path / "a" / rel / "best_model.pt").write_bytes(b"x")
    got = E.find_adapter_checkpoints(str(tmp_path / "a"))
    assert set(got) == {"L1", "L5F"} and "20260202" in got["L1"]
    os.makedirs(tmp_path / "b" / "L4_20260209_011428")
    (tmp_path / "b" / "L4_20260209_011428" / "best_model.pt").write_bytes(b"x")
    assert set(E.find_adapter_checkpoints(str(tmp_path / "b"))) == {"L4"}
    assert set(E.find_adapter_checkpoints(str(tmp_path / "b" / "L4_20260209_011428"))) == {"L4"}
    assert E.find_adapter_checkpoints(None) == {} and E.find_adapter_checkpoints(str(tmp_path / "zz")) == {}


def test_config_families_and_aggregate():
    cfgs = E.build_sd_configs(["L1", "L4", "B1", "L5F"])
    names = [c["name"] for c in cfgs]
    assert names == ["L1", "L4", "B1", "L5F", "L1+L5F", "L4+L5F", "L1+VL", "L4+VL"]
    assert [c["name"] for c in E.build_sd_configs(["L1", "B1"], only=["B1", "L1+VL"])] == ["B1", "L1+VL"]
    samples = [{"baseline": {"prefill_ms": 10.0, "decode_ms": 90.0, "total_ms": 100.0},
                "configs": {"L1": {"speedup": 2.0, "parallel_prefill_ms": 10.0, "sd_decode_ms": 40.0,
                                   "sd_total_ms": 50.0, "accept_rate": 0.5, "overlap_ms": 1.0, "lossless": True}}},
               {"baseline": {"prefill_ms": 20.0, "decode_ms": 80.0, "total_ms": 100.0},
                "configs": {"L1": {"speedup": 1.0, "parallel_prefill_ms": 20.0, "sd_decode_ms": 80.0,
                                   "sd_total_ms": 100.0, "accept_rate": 0.25, "overlap_ms": 3.0, "lossless": True,
                                   "num_free_tok": 4}}}]
    agg = E.aggregate(samples, cfgs)
    assert agg["VL baseline"]["prefill_ms_mean"] == 15.0 and agg["L1"]["speedup_mean"] == 1.5
    assert agg["L1"]["accept_rate_mean"] == 0.375 and agg["L1"]["lossless_all"] and "L4" not in agg
    st = E.stage_summary([{"vision_ms": 2.0, "prefill_ms": 10.0, "decode_ms": 100.0, "total_ms": 112.0,
                           "num_tokens": 50, "prefill_len": 600}])
    assert st["decode_tok_per_s"] == pytest.approx(500.0)


@pytest.mark.gpu
def test_e2e_benchmark_small_models(tmp_path, dev):
    out = tmp_path / "out"
    res = E.main(["--random_init", "small", "--max_samples", "2", "--warmup", "1", "--max_new_tokens", "16",
                  "--max_questions", "1", "--stage_samples", "1", "--output_dir", str(out)])
    assert res["n_measured"] == 2
    agg = res["aggregate"]
    for name in ("VL baseline", "L1", "L4", "B1", "L5F", "L1+L5F", "L1+VL"):
        assert name in agg, name
    for name, e in agg.items():
        if name != "VL baseline":
            assert e["lossless_all"], name
    assert agg["L1+VL"]["num_free_tok_mean"] >= 1
    files = os.listdir(out)
    assert any(f.endswith(".json") for f in files) and any(f.endswith(".md") for f in files)
    assert sum(f.endswith(".png") for f in files) == 4
    assert set(res["stage_timing"]) == {"EventGPT", "Video-LLaVA"}
