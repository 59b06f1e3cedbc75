"""Stage benchmark: SD arithmetic, CPU run on tiny models, GPU run on the engines."""
import os

import pytest
import torch

from eventgpt_amd.benchmarks import stages as S


def test_sd_arithmetic():
    assert S.calculate_acceptance_rate([1, 2, 3, 4], [1, 9, 3]) == (2 / 3, 2, 3)
    assert S.calculate_acceptance_rate([], [1]) == (0.0, 0, 0)
    # alpha 0.5, gamma 2: expected 1.75 tokens per step at cost 0.1*2+1
    assert S.calculate_speculative_speedup(0.1, 0.5, 2) == pytest.approx(1.75 / 1.2)
    assert S.calculate_speculative_speedup(0.0, 0.5, 2) == 0.0
    assert S.calculate_speculative_speedup(0.2, 1.0, 4) == pytest.approx((1 - 0.99 ** 5) / 0.01 / 1.8)

    class Shift(torch.nn.Module):
        def forward(self, ids):
            return torch.nn.functional.one_hot((ids + 1) % 10, 10).float()
    r = S.compute_aligned_acceptance_rate([1, 2, 3], [2, 3, 9], Shift())
    assert r["aligned_matched"] == 2 and r["aligned_acceptance_rate"] == pytest.approx(2 / 3)
    assert S.compute_aligned_acceptance_rate([1], [1], None)["method"] == "no_adapter"


def _check(res, tmp):
    assert set(res["summaries"]) == {"EventGPT", "Video-LLaVA"}
    e = res["summaries"]["EventGPT"]
    assert e["samples"] == 2 and e["stage3_avg"] > 0 and e["stage4_avg"] > 0 and e["decode_tps_avg"] > 0
    assert e["ttft_avg"] == pytest.approx(sum(e[f"stage{i}_avg"] for i in range(1, 5)), rel=1e-6)
    assert e["kv_cache_mb_avg"] > 0 and res["shapes"]["EventGPT"]["inputs_embeds"][0] == res["samples"]["EventGPT"][0]["prefill_tokens"]
    c = res["comparison"]
    assert c["c_ratio"] > 0 and "theoretical_sd_speedup" in c and 0 <= res["acceptance"]["acceptance_rate_mean"] <= 1
    files = os.listdir(tmp)
    assert any(f.endswith(".json") for f in files) and any(f.endswith(".md") for f in files)


def test_stage_benchmark_cpu(tmp_path):
    res = S.main(["--random_init", "tiny", "--device", "cpu", "--max_samples", "2", "--warmup", "1",
                  "--max_new_tokens", "6", "--speculative", "--from_npy", "--output_dir", str(tmp_path)])
    _check(res, tmp_path)
    assert res["config"]["from_npy"] and res["shapes"]["EventGPT"]["pixel_values"][0] == 5
    one = S.main(["--random_init", "tiny", "--device", "cpu", "--max_samples", "1", "--warmup", "0",
                  "--max_new_tokens", "4", "--compare_1vs8", "--eventgpt_only", "--output_dir", str(tmp_path)])
    assert one["shapes"]["EventGPT"]["pixel_values"][0] == 1 and "comparison" not in one


def test_llava15_grid_variant(tmp_path):
    """The single-image verifier variant sees one grid image: 1/T of the visual tokens."""
    res = S.main(["--random_init", "tiny", "--device", "cpu", "--max_samples", "1", "--warmup", "0",
                  "--max_new_tokens", "4", "--llava15_grid", "--output_dir", str(tmp_path)])
    vl, grid = res["summaries"]["Video-LLaVA"], res["summaries"]["LLaVA-1.5 (2x4 grid)"]
    sh_v, sh_g = res["shapes"]["Video-LLaVA"], res["shapes"]["LLaVA-1.5 (2x4 grid)"]
    T = sh_v["pixel_values_videos"][1]
    assert sh_g["pixel_values_videos"][1] == 1 and T > 1
    assert sh_g["video_features"][1] * T == sh_v["video_features"][1]
    assert grid["prefill_tokens_avg"] < vl["prefill_tokens_avg"]


def test_decoupling_check(tmp_path):
    res = S.main(["--random_init", "tiny", "--device", "cpu", "--max_samples", "1", "--warmup", "0",
                  "--max_new_tokens", "6", "--verify_decoupling", "--output_dir", str(tmp_path)])
    d = res["decoupling_check"]
    assert d["identical_tokens"] and d["n_tokens"] >= 1
    assert d["hooked"]["forwards"] == d["n_tokens"] or d["hooked"]["forwards"] == 6
    assert d["hooked"]["vision_s"] > 0 and d["hooked"]["prefill_s"] > 0
    assert 0.3 < d["hook_coverage"] <= 1.05
