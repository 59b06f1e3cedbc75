"""The inference flow on the event sample that ships with the reference
(``samples/sample1.npy``: inference.py:48-63, common/common.py:110-129).  The file is a pickled
dict; it is read symbolically (no opcode executed).  Skipped when the reference tree is absent.
Weights are random (no checkpoint is available offline), so what is pinned is everything up to
the model: event count and extent, the five rasters, the 582 event tokens, the prompt layout."""
import os

import numpy as np
import pytest
import torch

SAMPLE = "/root/reference/samples/sample1.npy"
needs_sample = pytest.mark.skipif(not os.path.exists(SAMPLE), reason="reference sample not present")
QUERY = "Describe in detail what happened in the scene."


@needs_sample
def test_sample_events_and_rasters():
    from eventgpt_amd.data import event_frames, load_event_npy
    from eventgpt_amd.data.events import rasterize_sequential, slice_bounds_by_count
    ev = load_event_npy(SAMPLE)
    n = len(ev["t"])
    assert n == 132_268 and ev["x"].dtype == np.uint16 and ev["p"].dtype == np.uint8
    assert int(ev["x"].max()) == 639 and int(ev["y"].max()) == 479 and int(ev["t"].max()) == 49_595
    assert bool((np.diff(ev["t"].astype(np.int64)) >= 0).all())           # time-sorted
    frames = event_frames(ev, 5)
    assert len(frames) == 5
    for f, (a, b) in zip(frames, slice_bounds_by_count(n, 5)):
        want = rasterize_sequential(ev["x"][a:b], ev["y"][a:b], ev["p"][a:b])   # the reference's per-event loop
        assert f.dtype == np.uint8 and np.array_equal(f, want)
        assert f.shape == (int(ev["y"][a:b].max()) + 1, int(ev["x"][a:b].max()) + 1, 3)
        coloured = (f != 255).any(-1)
        assert 0 < coloured.sum() <= b - a


@needs_sample
def test_inference_flow_on_the_sample_cpu():
    """inference.py's path with a small random decoder behind the real 336/14 tower geometry:
    5 frames x 577 tokens pool to 582 event tokens spliced in place of the one <event> id."""
    from eventgpt_amd.cli import inference as inf
    from eventgpt_amd.constants import EVENT_TOKEN_INDEX
    from eventgpt_amd.data import prepare_event_prompt, tokenizer_event_token
    args = inf.build_parser().parse_args(["--random_init", "stub125m", "--event_frame", SAMPLE, "--query", QUERY,
                                          "--temperature", "0", "--max_new_tokens", "4", "--device", "cpu"])
    model, tok, device, dtype = inf.load_model(args)
    model.eval()
    size, tensors = inf.process_event_data(SAMPLE, model.get_visual_tower().event_processor, device, dtype)
    assert len(tensors) == 5 and all(t.shape == (3, 336, 336) for t in tensors) and size == [480, 640]
    ids = tokenizer_event_token(prepare_event_prompt(QUERY, "eventgpt_v1"), tok, EVENT_TOKEN_INDEX,
                                return_tensors="pt").unsqueeze(0)
    assert int((ids == EVENT_TOKEN_INDEX).sum()) == 1
    with torch.no_grad():
        out = model.prepare_inputs_labels_for_multimodal(ids, None, torch.ones_like(ids), None, None,
                                                         event_tensors=tensors, event_image_sizes=size)
    embeds = out[4]
    assert embeds.shape[1] == ids.shape[1] - 1 + 582
    text = inf.run(["--random_init", "stub125m", "--event_frame", SAMPLE, "--query", QUERY, "--temperature", "0",
                    "--max_new_tokens", "4", "--device", "cpu"])
    assert isinstance(text, str)


def _matched_stream(path: str) -> str:
    """A stream with the sample's statistics (132 268 events, 640 x 480, 49 595 us, uint16 / uint8 fields,
    time-sorted) written as the same kind of file -- a pickled dict in a ``.npy`` -- and read back through
    the symbolic loader like the real one."""
    from eventgpt_amd.data.events import save_event_npy, synthetic_events
    ev = synthetic_events(132_268, 640, 480, 49_596, seed=1)
    save_event_npy(path, ev)
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["matched", "reference"])
def test_sample_on_the_gpu_front_end_and_engine(dev, source, tmp_path):
    """Device rasters == host rasters on the stream; the CLI flow runs through the engine.  ``reference``
    is the file that ships with the reference (absent on the GPU boxes: skipped there), ``matched`` a
    generated stream of the same size, extent, duration and file format."""
    from eventgpt_amd.cli import inference as inf
    from eventgpt_amd.data import load_event_npy
    from eventgpt_amd.data.events import rasterize, slice_bounds_by_count
    from eventgpt_amd.ops import event_ops as eo
    if source == "reference":
        if not os.path.exists(SAMPLE):
            pytest.skip("reference sample not present on this machine")
        path = SAMPLE
    else:
        path = _matched_stream(str(tmp_path / "matched.npy"))
    ev = load_event_npy(path)
    assert len(ev["t"]) == 132_268 and ev["x"].dtype == np.uint16 and ev["p"].dtype == np.uint8
    d = eo.events_to_device(ev, dev)
    img = eo.rasterize_gpu(d, 5, (480, 640)).cpu().numpy()
    for i, (a, b) in enumerate(slice_bounds_by_count(len(ev["t"]), 5)):
        assert np.array_equal(img[i], rasterize(ev["x"][a:b], ev["y"][a:b], ev["p"][a:b], size=(480, 640)))
    hist, _ = eo.voxel_grid_gpu(d, 5, (480, 640))
    assert int(hist.sum()) == len(ev["t"])
    text = inf.run(["--random_init", "stub125m", "--event_frame", path, "--query", QUERY, "--temperature", "0",
                    "--max_new_tokens", "8", "--device", "cuda"])
    assert isinstance(text, str)
