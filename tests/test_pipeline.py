"""Dataset access and the three extractors on tiny random models (CPU)."""
import json
import os

import pytest
import torch

from eventgpt_amd.align.store import load_chunked_hidden_states, read_index
from eventgpt_amd.pipeline import dataset as D
from eventgpt_amd.pipeline import extract_hidden_states as X
from eventgpt_amd.pipeline import extract_hidden_states_1f as X1
from eventgpt_amd.pipeline import extract_vl_lm_head as XL


@pytest.fixture(scope="module")
def data_dir(tmp_path_factory):
    return D.make_synthetic_dataset(str(tmp_path_factory.mktemp("ds")), n_samples=3, n_events=4000)


def test_dataset_layout(data_dir):
    ds = D.InstructionDataset(data_dir)
    assert len(ds) == 3 and len(ds.event_image_paths(ds[0])) == 5
    assert all(os.path.exists(p) for p in ds.event_image_paths(ds[1], "event_image_1f"))
    assert ds.event_npy_path(ds[0]).endswith("seq000.npy") and ds.video_source(ds[2]).endswith("seq002.npy")
    assert ds.question(ds[1]) == "What is moving?"
    qs = D.load_questions(os.path.join(data_dir, "top50_questions.json"), 1)
    assert qs == [D.DEFAULT_QUESTION] and D.load_questions(None) == [D.DEFAULT_QUESTION]
    a, b = D.shard_pairs(3, 2, 0, 2), D.shard_pairs(3, 2, 1, 2)
    assert sorted(a + b) == [(s, q) for s in range(3) for q in range(2)] and not set(a) & set(b)
    assert (0, 0) not in D.shard_pairs(3, 2, 0, 1, done={(0, 0)})


def test_extract_resume_and_1f(data_dir, tmp_path):
    out = str(tmp_path / "hs")
    base = ["--dataset_dir", data_dir, "--output_dir", out, "--random_init", "tiny", "--device", "cpu",
            "--max_new_tokens", "8", "--max_questions", "2", "--chunk_size", "2"]
    s1 = X.main(base + ["--max_samples", "2"])
    assert s1["new"] == 4 and s1["errors"] == 0
    store = os.path.join(out, "chunked_train_1s_bf16")
    with pytest.raises(SystemExit):
        X.main(base)
    s2 = X.main(base + ["--resume"])
    assert s2["new"] == 2 and s2["skipped"] == 4
    d = load_chunked_hidden_states(store)
    assert d["egpt_hidden"].shape[0] == 6 and d["egpt_hidden"].shape[2] == 64
    assert int(d["seq_lens"].min()) >= 5
    pairs = sorted((m["sample_idx"], m["question_idx"]) for m in d["metadata"])
    assert pairs == [(s, q) for s in range(3) for q in range(2)]
    assert all(k in d["metadata"][0] for k in ("question", "egpt_text", "vl_text", "seq_len"))
    # deterministic: the same pair extracted again gives the same states
    out1 = str(tmp_path / "hs1")
    st = X1.main(["--existing_chunks", store, "--dataset_dir", data_dir, "--output_dir", out1, "--random_init",
                  "tiny", "--device", "cpu", "--max_new_tokens", "8", "--max_questions", "2",
                  "--event_image_key", "event_image"])
    assert st["new"] == 6
    store1 = os.path.join(out1, "chunked_train_1s_bf16_event_image")
    d1 = load_chunked_hidden_states(store1)
    assert torch.allclose(d1["egpt_hidden"], d["egpt_hidden"], atol=1e-5)
    assert torch.equal(d1["vl_hidden"], d["vl_hidden"])
    idx = read_index(store1)
    assert [c["source_chunk"] for c in idx["chunks"]] == [c["path"] for c in read_index(store)["chunks"]]
    st2 = X1.main(["--existing_chunks", store, "--dataset_dir", data_dir, "--output_dir", out1, "--random_init",
                   "tiny", "--device", "cpu", "--max_new_tokens", "8", "--max_questions", "2",
                   "--event_image_key", "event_image", "--resume"])
    assert st2["new"] == 0 and st2["skipped_chunks"] == 3


def test_stop_flag_flushes(data_dir, tmp_path):
    from eventgpt_amd.align.store import ChunkedHiddenStateWriter

    class FakeExtractor:
        def __init__(self, stop):
            self.n, self.stop = 0, stop

        def extract_pair(self, ds, sample, q):
            self.n += 1
            if self.n == 2:
                self.stop.set = True       # "signal" arrives during the second pair
            h = torch.randn(6, 8)
            return {"egpt_hidden": h, "vl_hidden": h, "egpt_text": "", "vl_text": "", "seq_len": 6}

    stop = X.StopFlag()
    w = ChunkedHiddenStateWriter(str(tmp_path / "s"), chunk_size=100)
    stats = X.run_extraction(FakeExtractor(stop), D.InstructionDataset(data_dir), ["q"], w, stop=stop)
    assert stats["interrupted"] == 1 and stats["new"] == 2
    assert read_index(str(tmp_path / "s"))["total_samples"] == 2      # flushed despite chunk_size 100


def test_merge_rank_stores(tmp_path):
    from eventgpt_amd.align.store import ChunkedHiddenStateWriter
    for r in range(2):
        w = ChunkedHiddenStateWriter(str(tmp_path / f"rank_{r:02d}"), chunk_size=2, config={"hidden_dim": 4})
        for i in range(3):
            w.add_sample(torch.full((5, 4), float(r)), torch.zeros(5, 4), {"sample_idx": i, "question_idx": r})
        w.flush()
    X.merge_rank_stores(str(tmp_path))
    d = load_chunked_hidden_states(str(tmp_path))
    assert d["egpt_hidden"].shape == (6, 5, 4) and d["egpt_hidden"][:3].sum() == 0 and d["egpt_hidden"][3:].min() == 1


def test_lm_head_extractor(tmp_path):
    from eventgpt_amd.models.video_llava import VideoLlavaConfig, VideoLlavaModel
    m = VideoLlavaModel.from_config(VideoLlavaConfig.tiny(), torch.float32, "cpu", seed=4)
    m.save_pretrained(str(tmp_path / "vl"))
    out = XL.main(["--model_path", str(tmp_path / "vl"), "--output", str(tmp_path / "lm.pt")])
    blob = torch.load(out, weights_only=True)
    assert torch.equal(blob["lm_head_weight"], m.lm_head.weight.data) and blob["vocab_size"] == 320
    from eventgpt_amd.eval.acceptance import load_lm_head
    assert load_lm_head(out).shape == (320, 64)


def test_monitor_progress_wait_and_chain(tmp_path):
    import os as _os
    import subprocess as _sp
    import sys as _sys

    from eventgpt_amd.align.store import ChunkedHiddenStateWriter
    from eventgpt_amd.pipeline import monitor as M
    store = str(tmp_path / "store")
    assert M.extraction_progress(store)["exists"] is False
    w = ChunkedHiddenStateWriter(store, chunk_size=4)
    for i in range(10):
        w.add_sample(torch.randn(5, 8), torch.randn(5, 8), {"sample_idx": i, "question_idx": 0})
    w.flush()
    p = M.extraction_progress(store, expected=10)
    assert p["total_samples"] == 10 and p["n_chunks"] == 3 and p["complete"]
    assert "10 / 10 samples" in M.format_progress(p)
    # complete store: returns at once
    assert M.wait_for_extraction(store, 10, poll_s=0.01, out=lambda *_: None)["status"] == "complete"
    # short of the target with a watched process that has exited
    child = _sp.Popen([_sys.executable, "-c", "pass"])
    child.wait()
    r = M.wait_for_extraction(store, 50, pid=child.pid, poll_s=0.01, out=lambda *_: None)
    assert r["status"] == "extractor_exited" and r["total_samples"] == 10
    # short of the target, nothing watched: times out (sleep is injected, nothing really waits)
    naps = []
    r = M.wait_for_extraction(store, 50, poll_s=1.0, timeout_s=0.0, out=lambda *_: None, sleep=naps.append)
    assert r["status"] == "timeout"
    assert M.pid_alive(_os.getpid()) and not M.pid_alive(child.pid)
    # the chain stops at the first failing stage
    calls = []

    def fake_runner(cmd, log, out):
        calls.append(cmd)
        return 0 if "--adapter_level" in cmd and cmd[cmd.index("--adapter_level") + 1] == "1" else 3
    res = M.auto_train(store, None, [1, 2, 3], str(tmp_path / "tasks"), runner=fake_runner, out=lambda *_: None)
    assert [r["level"] for r in res] == ["L1", "L2"] and res[1]["train_rc"] == 3 and len(calls) == 2
    # and really trains + evaluates one level end to end on the CPU
    res = M.auto_train(store, store, [1], str(tmp_path / "tasks"), str(tmp_path / "logs"), num_epochs=1,
                       batch_size=4, extra_train_args=["--hidden_dim", "8", "--bottleneck_dim", "4"], cpu=True,
                       out=lambda *_: None)
    assert res[0]["train_rc"] == 0 and res[0]["eval_rc"] == 0, open(tmp_path / "logs" / "train_L1.log").read()[-800:]
    assert _os.path.exists(_os.path.join(res[0]["run_dir"], "eval", "acceptance_metrics.json"))
