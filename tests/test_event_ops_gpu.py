"""GPU event front end vs the numpy implementation."""
import numpy as np
import pytest
import torch

from eventgpt_amd.data import clip_preprocess as cp
from eventgpt_amd.data.events import (polarity_histogram, rasterize, slice_bounds_by_count,
                                      synthetic_events, voxel_grid)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conflict", ["last", "negative"])
def test_raster_matches_numpy(dev, conflict):
    from eventgpt_amd.ops import event_ops as eo
    ev = synthetic_events(50_000, 160, 120, seed=4)
    d = eo.events_to_device(ev, dev)
    img = eo.rasterize_gpu(d, 5, (120, 160), conflict).cpu().numpy()
    for i, (a, b) in enumerate(slice_bounds_by_count(50_000, 5)):
        want = rasterize(ev["x"][a:b], ev["y"][a:b], ev["p"][a:b], size=(120, 160), conflict=conflict)
        assert np.array_equal(img[i], want), i


def test_voxel_and_histogram_match_numpy(dev):
    from eventgpt_amd.ops import event_ops as eo
    ev = synthetic_events(30_000, 64, 48, seed=2)
    d = eo.events_to_device(ev, dev)
    hist, vox = eo.voxel_grid_gpu(d, 5, (48, 64))
    assert np.array_equal(hist.cpu().numpy(), polarity_histogram(ev, 5, (48, 64)))
    assert np.allclose(vox.cpu().numpy(), voxel_grid(ev, 5, (48, 64)))
    _, voxc = eo.voxel_grid_gpu(d, 5, (48, 64), signed=False, by="count")
    assert np.allclose(voxc.cpu().numpy(), voxel_grid(ev, 5, (48, 64), signed=False, by="count"))


@pytest.mark.parametrize("n,hw,bins", [(200_000, (480, 640), 5), (70_001, (97, 130), 3), (5000, (48, 64), 7)])
@pytest.mark.parametrize("by", ["time", "count"])
def test_voxel_lds_bands_equal_atomic_kernel_and_numpy(dev, n, hw, bins, by):
    """The per-band LDS histogram and the one-atomic-per-event kernel give identical counts;
    both match the host implementation, with out-of-canvas events dropped."""
    from eventgpt_amd.ops import event_ops as eo
    ev = synthetic_events(n, hw[1] + 6, hw[0] + 4, seed=n % 97)      # some events fall outside the canvas
    d = eo.events_to_device(ev, dev)
    h_lds, v_lds = eo.voxel_grid_gpu(d, bins, hw, by=by, privatised=True)
    h_at, v_at = eo.voxel_grid_gpu(d, bins, hw, by=by, privatised=False)
    assert torch.equal(h_lds, h_at) and torch.equal(v_lds, v_at)
    inside = (ev["x"] < hw[1]) & (ev["y"] < hw[0])
    assert int(h_lds.sum()) == int(inside.sum())
    if by == "time":
        keep = {k: v[inside] for k, v in ev.items()}
        t0, t1 = int(ev["t"].min()), int(ev["t"].max())
        b = np.clip(((keep["t"] - t0).astype(np.float64) / max(t1 - t0, 1) * bins).astype(np.int64), 0, bins - 1)
        want = np.zeros((bins, 2) + hw, np.int64)
        np.add.at(want, (b, (keep["p"] != 0).astype(np.int64), keep["y"], keep["x"]), 1)
        assert np.array_equal(h_lds.cpu().numpy(), want)


def test_voxel_unsorted_stream_takes_the_atomic_kernel(dev):
    from eventgpt_amd.ops import event_ops as eo
    ev = synthetic_events(20_000, 64, 48, seed=5)
    perm = np.random.default_rng(0).permutation(20_000)
    shuffled = {k: v[perm] for k, v in ev.items()}
    d = eo.events_to_device(shuffled, dev)
    assert eo._bin_bounds(d, 5, "time", int(ev["t"].min()), int(ev["t"].max() - ev["t"].min())) is None
    hist, _ = eo.voxel_grid_gpu(d, 5, (48, 64), privatised=True)
    assert np.array_equal(hist.cpu().numpy(), polarity_histogram(ev, 5, (48, 64)))


@pytest.mark.parametrize("hw", [(480, 640), (120, 160), (336, 336), (400, 300)])
def test_clip_preprocess_matches_host(dev, hw):
    from eventgpt_amd.ops import event_ops as eo
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (2, hw[0], hw[1], 3), dtype=np.uint8)
    got = eo.clip_preprocess_gpu(torch.from_numpy(frames).to(dev)).float().cpu()
    want = torch.stack([torch.from_numpy(cp.preprocess_image(f)) for f in frames])
    assert got.shape == (2, 3, 336, 336)
    # bf16 output: compare after the same rounding
    assert torch.equal(got, want.to(torch.bfloat16).float())


def test_event_frames_gpu_end_to_end(dev):
    from eventgpt_amd.ops import event_ops as eo
    from eventgpt_amd.data.events import event_frames
    ev = synthetic_events(60_000, 640, 480, seed=9)
    size, px = eo.event_frames_gpu(ev, dev, 5)
    assert size == [480, 640] and px.shape == (5, 3, 336, 336) and px.dtype == torch.bfloat16
    host = event_frames(ev, 5)
    same_canvas = [f for f in host if f.shape[:2] == (480, 640)]
    if len(same_canvas) == 5:
        want = torch.stack([torch.from_numpy(cp.preprocess_image(f)) for f in host])
        assert torch.equal(px.float().cpu(), want.to(torch.bfloat16).float())


def test_preprocess_frames_device_dispatch(dev):
    """The frame preprocessing entry point gives the same pixels from the GPU kernel
    and from the host path, and the verifier's processor uses it when given a device."""
    from eventgpt_amd.ops import event_ops as eo
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, (120, 160, 3), dtype=np.uint8) for _ in range(4)]
    g = eo.preprocess_frames(frames, dev, 56)
    c = eo.preprocess_frames(frames, "cpu", 56)
    assert g.is_cuda and g.shape == (4, 3, 56, 56) and g.dtype == torch.bfloat16
    assert torch.equal(g.cpu(), c)
    # unequal sizes (per-slice canvases) run the kernel frame by frame: same pixels as the host path
    mixed = frames[:2] + [rng.integers(0, 256, (100, 90, 3), dtype=np.uint8)]
    m = eo.preprocess_frames(mixed, dev, 56)
    assert m.is_cuda and m.shape == (3, 3, 56, 56)
    assert torch.equal(m.cpu(), eo.preprocess_frames(mixed, "cpu", 56))
    from eventgpt_amd.data.tokenizer import ByteTokenizer
    from eventgpt_amd.models.video_llava import VideoLlavaConfig, VideoProcessor
    cfg = VideoLlavaConfig.tiny()
    proc = VideoProcessor(cfg, ByteTokenizer())
    S = cfg.vision_config.image_size
    out = proc("USER: <video>\nwhat? ASSISTANT:", frames, device=dev)
    host = proc("USER: <video>\nwhat? ASSISTANT:", frames)
    assert out["pixel_values_videos"].shape == (1, 4, 3, S, S) and out["pixel_values_videos"].is_cuda
    assert torch.equal(out["pixel_values_videos"].float().cpu(), host["pixel_values_videos"].to(torch.bfloat16).float())
