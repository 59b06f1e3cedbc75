"""GPU event front end: events -> raster frames -> CLIP-normalised bf16 pixels.

Replaces the reference's host pipeline (python loop over every event to paint
the raster, PIL resize, numpy normalise — common/common.py:17-37,64-74,110-129)
with three launches on device-resident event arrays:

    event_raster    one atomic per event (last-wins / negative-wins), all T slices
    clip_preprocess PIL-exact two-pass bicubic + crop + normalise, all T frames
    (event_voxel    time-binned polarity histogram / signed voxel grid; LDS-privatised per band)

The resample tables are the ones ``data.clip_preprocess`` builds, so device and
host produce bit-identical uint8 images.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ..constants import CLIP_IMAGE_SIZE, CLIP_MEAN, CLIP_STD
from ..data import clip_preprocess as cp
from ..data.events import canvas_size

_TABLE_CACHE: Dict[tuple, tuple] = {}


def events_to_device(events: dict, device) -> Dict[str, torch.Tensor]:
    """int32 x/y/p and int64 t tensors on ``device`` (one H2D copy each)."""
    out = {}
    for k in ("x", "y", "p"):
        out[k] = torch.from_numpy(np.ascontiguousarray(events[k]).astype(np.int32)).to(device)
    out["t"] = torch.from_numpy(np.ascontiguousarray(events["t"]).astype(np.int64)).to(device)
    return out


def rasterize_gpu(ev: Dict[str, torch.Tensor], n_slices: int = 5,
                  size: Optional[Tuple[int, int]] = None, conflict: str = "last") -> torch.Tensor:
    """``uint8 [T, H, W, 3]``.  Every slice is painted on a canvas of the given (or
    whole-stream) extent.  NOTE: the host path sizes each slice's canvas from that
    slice's own max coordinates; pass ``size`` to pin it."""
    from . import hip
    if size is None:
        size = (int(ev["y"].max()) + 1, int(ev["x"].max()) + 1)
    mode = {"last": 0, "negative": 1}[conflict]
    return hip.raw.event_raster(ev["x"], ev["y"], ev["p"], n_slices, size[0], size[1], mode)


def _bin_bounds(ev: Dict[str, torch.Tensor], bins: int, by: str, t0: int, span: int) -> Optional[torch.Tensor]:
    """Device int32 ``[bins+1]``: events ``bounds[b]:bounds[b+1]`` form bin ``b`` -- or ``None``
    when binning by time on a stream that is not time-sorted (bins are then no index ranges)."""
    n = ev["t"].numel()
    dev = ev["t"].device
    if by == "count":
        # event i goes to bin floor(i * bins / n): bin b starts at ceil(b * n / bins)
        b = torch.arange(bins + 1, dtype=torch.int64)
        return ((b * n + bins - 1) // bins).to(torch.int32).to(dev)
    t = ev["t"]
    if n > 1 and not bool((t[1:] >= t[:-1]).all()):
        return None
    # first timestamp of every bin under the kernel's own arithmetic (IEEE double, same operation
    # order): bin(t) = clamp(int((t - t0) / span * bins)); start from the real-valued edge and
    # step to the exact integer boundary
    def which(ts: int) -> int:
        return min(max(int(float(ts - t0) / float(span) * bins), 0), bins - 1)
    edges = [t0]
    for b in range(1, bins):
        e = t0 + (b * span) // bins
        while which(e) >= b and e > t0:
            e -= 1
        while which(e) < b:
            e += 1
        edges.append(e)
    first = torch.searchsorted(t, torch.tensor(edges, dtype=t.dtype, device=dev))
    first[0] = 0
    return torch.cat([first, first.new_tensor([n])]).to(torch.int32)


def voxel_grid_gpu(ev: Dict[str, torch.Tensor], bins: int = 5,
                   size: Optional[Tuple[int, int]] = None, signed: bool = True,
                   by: str = "time", privatised: Optional[bool] = None,
                   chunks: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(hist int32 [bins,2,H,W], voxel fp32 [bins,H,W])``.

    Two kernels produce the same numbers: one global atomic per event, and (``privatised``,
    the default whenever the bins are index ranges: binning by count, or by time on a sorted
    stream) per-band counting in LDS with one global atomic per touched pixel."""
    from . import hip
    if size is None:
        size = (int(ev["y"].max()) + 1, int(ev["x"].max()) + 1)
    t0 = int(ev["t"].min()) if ev["t"].numel() else 0
    span = max(int(ev["t"].max()) - t0, 1) if ev["t"].numel() else 1
    if privatised is None:
        privatised = ev["t"].numel() >= 4096
    if privatised:
        bounds = _bin_bounds(ev, bins, by, t0, span)
        if bounds is not None:
            x, y, p = (a if a.data_ptr() % 16 == 0 else a.clone() for a in (ev["x"], ev["y"], ev["p"]))
            return hip.raw.event_voxel_ranges(x, y, p, bounds, size[0], size[1], signed, chunks)
    return hip.raw.event_voxel(ev["x"], ev["y"], ev["p"], ev["t"], t0, span, bins, size[0],
                               size[1], signed, by == "count")


_AXIS_CACHE: Dict[tuple, tuple] = {}


def _axis_table(n_in: int, n_out: int, device) -> tuple:
    """Device copies of the resample table of one axis (start, count, weight), cached per
    ``(n_in, n_out)``: clips whose rasters differ in one extent still share the other axis."""
    key = (n_in, n_out, str(device))
    hit = _AXIS_CACHE.get(key)
    if hit is None:
        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)
        if n_out != n_in:
            t = cp.resample_table(n_in, n_out)
            hit = (dev(t.start), dev(t.count), dev(t.weight))
        else:
            hit = (dev(np.zeros(1)), dev(np.zeros(1)), dev(np.zeros((1, 1))))
        _AXIS_CACHE[key] = hit
    return hit


def _tables(h: int, w: int, size: int, device) -> tuple:
    key = (h, w, size, str(device))
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        return hit
    rh, rw = cp.resized_shape(h, w, size)
    top, left = cp.center_crop_offsets(rh, rw, size)
    hs, hc, hw = _axis_table(w, rw, device)
    vs, vc, vw = _axis_table(h, rh, device)
    out = (hs, hc, hw, vs, vc, vw, rh, rw, top, left)
    _TABLE_CACHE[key] = out
    return out


def clip_preprocess_gpu(frames_u8: torch.Tensor, size: int = CLIP_IMAGE_SIZE) -> torch.Tensor:
    """``uint8 [T,H,W,3]`` on the GPU -> ``bf16 [T,3,size,size]``."""
    from . import hip
    t, h, w, _ = frames_u8.shape
    hs, hc, hw, vs, vc, vw, rh, rw, top, left = _tables(h, w, size, frames_u8.device)
    if rh < size or rw < size:
        raise ValueError("image too small to crop after resize")
    return hip.raw.clip_preprocess(frames_u8.contiguous(), hs, hc, hw, vs, vc, vw, size, rh, rw,
                                   top, left, list(CLIP_MEAN), list(CLIP_STD))


def preprocess_frames(frames, device, size: int = CLIP_IMAGE_SIZE, dtype=torch.bfloat16) -> torch.Tensor:
    """RGB uint8 frames -> normalised ``[T,3,size,size]`` on ``device``.
    On a GPU the frames are uploaded as uint8 (12x fewer bytes than fp32 pixels) and
    resized/cropped/normalised by one kernel; on the CPU the PIL-exact numpy path runs.
    Both produce the same pixels."""
    device = torch.device(device)
    arrs = [np.asarray(f) for f in frames]
    if device.type == "cuda" and all(a.dtype == np.uint8 and a.ndim == 3 for a in arrs):
        if len({a.shape for a in arrs}) == 1:
            u8 = torch.from_numpy(np.ascontiguousarray(np.stack(arrs))).to(device, non_blocking=False)
            return clip_preprocess_gpu(u8, size).to(dtype)
        # rasters of one clip can differ in size (the canvas follows each slice's extent):
        # one kernel call per frame, still two orders of magnitude faster than the host path
        # (np.array copies: image decoders hand out read-only buffers, which from_numpy warns about)
        outs = [clip_preprocess_gpu(torch.from_numpy(np.array(a, order="C"))[None].to(device), size)
                for a in arrs]
        return torch.cat(outs).to(dtype)
    px = [torch.from_numpy(cp.preprocess_image(a, size)) for a in arrs]
    return torch.stack(px).to(device, dtype)


def event_frames_gpu(events: dict, device, n_frames: int = 5, size: int = CLIP_IMAGE_SIZE,
                     conflict: str = "last") -> Tuple[list, torch.Tensor]:
    """Host event dict -> ``([H, W], bf16 [T,3,size,size])`` entirely on the GPU."""
    ev = events_to_device(events, device)
    hw = canvas_size(np.asarray(events["x"]), np.asarray(events["y"]))
    img = rasterize_gpu(ev, n_frames, hw, conflict)
    return list(hw), clip_preprocess_gpu(img, size)
