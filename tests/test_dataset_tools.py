"""DSEC recording access, clip building, raster preprocessing, video/labels/analysis tools (CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from eventgpt_amd.data import dsec
from eventgpt_amd.data.events import load_event_npy, synthetic_events
from eventgpt_amd.pipeline import dataset_tools as DT
from eventgpt_amd.pipeline.dataset import InstructionDataset


def _recording(tmp, name="zurich_00", seconds=3.0, n=60000, offset=5_000_000, seed=0):
    from PIL import Image
    seq = tmp / name
    rng = np.random.default_rng(seed)
    t = np.sort(rng.integers(0, int(seconds * 1e6), n)).astype(np.int64)
    ev = {"x": rng.integers(0, 64, n), "y": rng.integers(0, 48, n), "t": t, "p": rng.integers(0, 2, n)}
    dsec.save_recording(str(seq / "events" / "left" / "events.npz"), ev, t_offset=offset)
    img_dir = seq / "images" / "left" / "rectified"
    img_dir.mkdir(parents=True)
    stamps = np.arange(offset, offset + int(seconds * 1e6) + 1, 250_000)
    for i in range(len(stamps)):
        Image.fromarray(rng.integers(0, 255, (48, 64, 3), dtype=np.uint8)).save(img_dir / f"{i:06d}.png")
    np.savetxt(seq / "images" / "timestamps.txt", stamps, fmt="%d")
    return seq, ev, offset


def test_event_file_slicing(tmp_path):
    seq, ev, off = _recording(tmp_path)
    path = str(seq / "events" / "left" / "events.npz")
    assert dsec.get_num_events(path) == 60000
    with dsec.EventFile(path) as f:
        assert f.time_range == (int(ev["t"][0]) + off, int(ev["t"][-1]) + off)
        for a, b in ((off + 123_456, off + 1_000_001), (off, off + 10), (off + 2_999_000, off + 9_000_000)):
            got = f.by_timewindow(a, b)
            mask = (ev["t"] + off >= a) & (ev["t"] + off < b)
            assert np.array_equal(got["t"], ev["t"][mask] + off) and np.array_equal(got["x"], ev["x"][mask])
        ms = f.by_timewindow(off + 100_500, off + 200_900, exact=False)        # cut at whole milliseconds
        mask = (ev["t"] >= 100_000) & (ev["t"] < 200_000)
        assert np.array_equal(ms["t"], ev["t"][mask] + off)
        sl = f.by_index(10, 20)
        assert np.array_equal(sl["p"], ev["p"][10:20]) and sl["t"].dtype == np.int64
    assert np.array_equal(dsec.extract_from_h5_by_index(path, 5, 9)["y"], ev["y"][5:9])
    assert np.array_equal(dsec.extract_from_h5_by_timewindow(path, off + 100_500, off + 200_900)["t"], ms["t"])
    m2i = dsec.build_ms_to_idx(ev["t"])
    assert m2i[0] == 0 and m2i[7] == np.searchsorted(ev["t"], 7000) and m2i[-1] == len(ev["t"])
    with pytest.raises(ValueError):
        dsec.save_recording(str(tmp_path / "bad.npz"), {**ev, "t": ev["t"][::-1]})
    d = dsec.DSECDirectory(seq)
    assert d.events.event_file.name == "events.npz" and len(d.images.image_files_rectified) == 13
    assert d.images.timestamps[0] == off
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    (tmp_path / "a" / "f.txt").write_text("1"); (tmp_path / "b" / "f.txt").write_text("1")
    assert dsec.compare_dirs(tmp_path / "a", tmp_path / "b")
    (tmp_path / "b" / "f.txt").write_text("2")
    assert not dsec.compare_dirs(tmp_path / "a", tmp_path / "b")
    (tmp_path / "c.yaml").write_text("a: 1\nb: [2, 3]\n")
    assert dsec.yaml_file_to_dict(tmp_path / "c.yaml") == {"a": 1, "b": [2, 3]}


def test_clip_builder_and_rasters(tmp_path):
    assert DT.format_duration(5_000_000) == "5s" and DT.format_duration(200_000) == "200ms"
    assert DT.parse_duration_from_dir("/x/my_egpt_dsec_seq_500ms") == 0.5
    assert DT.sliding_clips(0, 3_000_000, 1_000_000) == [(i * 500_000, i * 500_000 + 1_000_000) for i in range(5)]
    assert DT.calculate_potential_clips(3_000_000) == {1_000_000: 5, 5_000_000: 0, 10_000_000: 0}
    seq, ev, off = _recording(tmp_path)
    out = str(tmp_path / "my_egpt_dsec_seq_1s")
    entries = DT.build_clips([str(seq)], out, 1_000_000)
    assert len(entries) == 5 and entries[1]["time_start"] == off + 500_000
    clip = load_event_npy(os.path.join(out, "event_npy", entries[1]["event_data"]), trusted=True)
    mask = (ev["t"] + off >= entries[1]["time_start"]) & (ev["t"] + off < entries[1]["time_end"])
    assert clip["t"].size == int(mask.sum()) == entries[1]["num_events"]
    assert len(os.listdir(os.path.join(out, "video", "zurich_00", "000001"))) == 4       # frames at 250 ms
    st = DT.preprocess_event_images(out, num_workers=2, num_frames=5)
    assert st == {"processed": 5, "skipped": 0, "failed": 0, "num_frames": 5, "json_key": "event_image"}
    assert DT.preprocess_event_images(out, num_workers=1, num_frames=5)["skipped"] == 5
    assert DT.preprocess_event_images(out, num_workers=1, num_frames=1)["json_key"] == "event_image_1f"
    assert DT.preprocess_event_images(out, num_workers=1, hz=2)["num_frames"] == 2
    ds = InstructionDataset(out)
    assert len(ds[0]["event_image"]) == 5 and len(ds[0]["event_image_1f"]) == 1 and len(ds[0]["event_image_2Hz"]) == 2
    assert all(os.path.exists(p) for p in ds.event_image_paths(ds[0]))
    from PIL import Image
    im = np.asarray(Image.open(ds.event_image_paths(ds[0], "event_image_1f")[0]))
    assert im.shape[2] == 3 and {tuple(c) for c in im.reshape(-1, 3)} <= {(255, 255, 255), (255, 0, 0), (0, 0, 255)}
    v = DT.preprocess_videos(out)
    assert v["mp4"] + v["npy"] == 5 and ds.video_source(ds[0]) is not None
    calls = []

    def answer(sample, q):
        calls.append((sample["id"], q))
        return f"answer to {q}", 0.9 if "moving" in q else 0.2
    qs = ["What is moving?", "What colour is the sky?"]
    r = DT.generate_answers(out, qs, answer, min_confidence=0.5)
    assert r == {"labelled": 5, "skipped": 0, "low_confidence": 5} and len(calls) == 10
    ds = InstructionDataset(out)
    assert ds[0]["conversations"][1]["value"] == "answer to What is moving?" and ds[0]["answer_confidence"] == [0.9]
    assert DT.generate_answers(out, qs[:1], answer)["skipped"] == 5
    rep = DT.analyze_dataset(out)
    assert rep["num_event_files"] == 5 and rep["top_questions"][0] == {"question": "What is moving?", "count": 5}
    assert 0.9 < rep["duration_s"]["mean"] <= 1.0 and rep["event_rate_per_s"] == pytest.approx(20000, rel=0.05)
    assert json.load(open(os.path.join(out, "top50_questions.json")))[0]["count"] == 5


def test_alignment_feature_builder(tmp_path):
    from eventgpt_amd.data.tokenizer import ByteTokenizer
    from eventgpt_amd.models.video_llava import VideoLlavaConfig, VideoLlavaModel, VideoProcessor
    from eventgpt_amd.pipeline.dataset import make_synthetic_dataset
    data = make_synthetic_dataset(str(tmp_path / "ds"), n_samples=2, n_events=2000)
    cfg = VideoLlavaConfig.tiny()
    model = VideoLlavaModel.from_config(cfg, torch.float32, "cpu", seed=0)
    proc = VideoProcessor(cfg, ByteTokenizer())
    ds = InstructionDataset(data)
    r = DT.build_alignment_features(model, proc, ds, str(tmp_path / "feat"), max_new_tokens=4)
    assert r == {"written": 2, "skipped": 0}
    blob = torch.load(tmp_path / "feat" / "000001.pt", weights_only=True)
    assert blob["vision_features"].shape == (cfg.num_frames, 64) and blob["tokens"].shape == (4,)
    assert DT.build_alignment_features(model, proc, ds, str(tmp_path / "feat"))["skipped"] == 2
    ans = DT.model_answerer(model, proc, ds, max_new_tokens=3)(ds[0], "What?")
    assert isinstance(ans[0], str) and 0.0 < ans[1] <= 1.0


def test_snapshot_io_and_validation(tmp_path):
    import json as _json
    import os as _os

    import numpy as _np

    from eventgpt_amd.pipeline import dataset_io as DIO
    from eventgpt_amd.pipeline.dataset import JSON_NAME, make_synthetic_dataset
    root = make_synthetic_dataset(str(tmp_path / "ds"), n_samples=4, n_events=500, width=32, height=24)
    entries = DIO.load_snapshot(root)
    assert len(entries) == 4 and DIO.find_snapshot_json(root).endswith(JSON_NAME)
    rep = DIO.validate_dataset(root, n_event_checks=-1, out=lambda *_: None)
    assert rep["n_valid"] == 4 and not rep["issues"] and len(rep["event_stats"]) == 4
    assert rep["event_stats"][0]["num_events"] == 500 and rep["event_stats"][0]["width"] <= 32
    # break three entries in three different ways
    entries[0]["conversations"][0]["from"] = "gpt"
    entries[1].pop("event_data")
    _os.remove(_os.path.join(root, "event_npy", entries[2]["event_data"]))
    with open(_os.path.join(root, JSON_NAME), "w") as f:
        _json.dump(entries, f)
    # and make the fourth event file something the array-only loader must refuse
    class Evil:
        def __reduce__(self):
            return (print, ("executed",))
    _np.save(_os.path.join(root, "event_npy", entries[3]["event_data"]), _np.array({"x": Evil()}, dtype=object),
             allow_pickle=True)
    rep = DIO.validate_dataset(root, n_event_checks=-1, out=lambda *_: None)
    assert rep["n_valid"] == 0 and len(rep["issues"]) == 4
    kinds = " | ".join(i["issue"] for i in rep["issues"])
    assert "human -> gpt" in kinds and "missing fields" in kinds and "not found" in kinds and "event data" in kinds
    ds = DIO.load_snapshot(root, as_hf=True)
    assert len(ds) == 4 and "conversations" in ds.column_names
