"""IMU 5-stage benchmark and the causal-LM pre-training script on tiny configs (CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from eventgpt_amd.benchmarks import imu as I
from eventgpt_amd.cli import train_lm as TL


def test_imu_data_and_models():
    d = I.generate_synthetic_imu_data(3, 120)
    assert len(d) == 3 and d[0].shape == (120, 6) and abs(d[0][:, 2].mean() - 9.8) < 0.2
    n = I.normalize_imu_data(d[0])
    assert np.allclose(n.mean(0), 0, atol=1e-5) and np.allclose(n.std(0), 1, atol=1e-3)
    e = I.generate_synthetic_imu_ego4d(2, 1.0)
    assert e[0]["signal"].shape == (6, 400)
    assert I.pad_imu_signal(torch.ones(6, 10), 16).shape == (6, 16) and I.pad_imu_signal(torch.ones(6, 30), 16).shape == (6, 16)
    m = I.ImuChatModel.random_init("llasa", "tiny")
    assert m.encode_imu(torch.randn(1, 120, 6)).shape == (1, 120, 64)
    m2 = I.ImuChatModel.random_init("onellm", "tiny")
    assert m2.encode_imu(torch.randn(1, 6, 2000)).shape == (1, 30, 64)


def test_imu_benchmark_cli(tmp_path):
    out = I.main(["--size", "tiny", "--device", "cpu", "--max_samples", "2", "--max_new_tokens", "5",
                  "--output_dir", str(tmp_path)])
    assert set(out["summaries"]) == {"LLaSA-shaped", "OneLLM-shaped"}
    s = out["summaries"]["LLaSA-shaped"]
    assert s["samples"] == 2 and s["stage3_avg"] > 0 and s["decode_tokens_avg"] >= 1
    assert out["samples"]["OneLLM-shaped"][0]["shapes"]["imu_tokens"] == [1, 30, 64]
    assert any(f.startswith("imu_5stages_") for f in os.listdir(tmp_path))


def test_event_tokenizer_and_lm_training(tmp_path):
    ev = {"x": np.array([0, 319, 160]), "y": np.array([0, 239, 120]), "t": np.array([0, 1, 1000]),
          "p": np.array([0, 1, 1])}
    tok = TL.tokenize_events(ev, 320, 240)
    # event 0: p0, dt 0 -> bucket 0, cell (0,0); event 1: p1, dt 1 -> bucket 1, cell (15,15)
    assert tok[0] == 3 and tok[1] == 3 + ((1 * TL.TIME_BINS + 1) * TL.GRID + 15) * TL.GRID + 15
    assert tok.max() < TL.VOCAB and tok[2] == 3 + ((TL.TIME_BINS + 7) * TL.GRID + 8) * TL.GRID + 8
    res = TL.main(["--synthetic", "2", "--output_dir", str(tmp_path / "lm"), "--block_size", "64", "--layers", "1",
                   "--hidden", "64", "--num_train_epochs", "2", "--per_device_train_batch_size", "16",
                   "--gradient_accumulation_steps", "2", "--learning_rate", "3e-3", "--warmup_steps", "2",
                   "--eval_steps", "5", "--save_steps", "5", "--logging_steps", "1000", "--cpu"])
    assert res["steps"] > 5 and res["log"] and res["log"][-1]["eval_loss"] < res["log"][0]["eval_loss"] + 1e-3
    assert res["final_eval_loss"] < np.log(TL.VOCAB)
    for f in ("best_model.pt", "final_model.pt", "train_log.json", "trainer_state_rank0.pt"):
        assert os.path.exists(tmp_path / "lm" / f), f
    again = TL.main(["--synthetic", "2", "--output_dir", str(tmp_path / "lm"), "--block_size", "64", "--layers", "1",
                     "--hidden", "64", "--num_train_epochs", "2", "--per_device_train_batch_size", "16",
                     "--gradient_accumulation_steps", "2", "--eval_steps", "5", "--save_steps", "5", "--cpu",
                     "--resume"])
    assert again["steps"] >= res["steps"] - 5
