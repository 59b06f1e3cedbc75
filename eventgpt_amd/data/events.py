"""Event-stream IO, slicing, rasterisation and voxelisation (host side).

An event stream is a dict of four equal-length 1-D arrays ``x, y, t, p``.
This module is the numpy oracle and the CPU path; the GPU path lives in
``eventgpt_amd.ops.event_ops`` (HIP kernels) and is tested against it.

Behaviour parity (reference: common/common.py:17-37, 64-107):
  * ``split_by_count`` cuts the stream into ``n`` slices of equal event count,
    the last slice taking the remainder.
  * ``rasterize`` paints a white canvas of size ``(y.max()+1, x.max()+1)``;
    polarity 0 → blue, otherwise red.  Two conflict policies are offered:
    ``"last"`` (the latest event at a pixel wins — the reference's sequential
    loop) and ``"negative"`` (a negative event anywhere at the pixel wins — the
    reference's vectorised dataset preprocessor,
    feasible/my_egpt_dsec_dataset/preprocess_event_images.py:58-90).
  * ``split_by_time`` groups events into fixed time bins (50 ms default).
"""
from __future__ import annotations

import io
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

EventDict = Dict[str, np.ndarray]

_RED = np.array([255, 0, 0], dtype=np.uint8)
_BLUE = np.array([0, 0, 255], dtype=np.uint8)


# ----------------------------------------------------------------------- IO
class _Sym:
    """Node of a pickle read *symbolically*: a named global, a call of one, or an object with state."""

    __slots__ = ("kind", "a", "b")

    def __init__(self, kind, a=None, b=None):
        self.kind, self.a, self.b = kind, a, b


_MARK = object()


def _read_pickle_symbolic(f):
    """Walk a pickle stream opcode by opcode WITHOUT executing it: globals stay names, ``REDUCE`` and
    ``BUILD`` only record (callable name, arguments, state).  Nothing is imported, called or
    constructed, so a hostile file can at most produce a tree this module then refuses.  Only the
    opcodes numpy's array pickles use are understood; any other opcode raises ``ValueError``.

    Event samples in the wild are ``np.save``-d python dicts (reference: common/common.py:111-112),
    i.e. pickles; this is how they are read when the file is not trusted."""
    import pickletools
    stack, memo = [], {}

    def pop_mark():
        k = len(stack) - 1
        while stack[k] is not _MARK:
            k -= 1
        items = stack[k + 1:]
        del stack[k:]
        return items

    for op, arg, _pos in pickletools.genops(f):
        n = op.name
        if n in ("PROTO", "FRAME"):
            pass
        elif n == "GLOBAL":
            mod, name = arg.split(" ")
            stack.append(_Sym("global", mod, name))
        elif n == "STACK_GLOBAL":
            name, mod = stack.pop(), stack.pop()
            stack.append(_Sym("global", mod, name))
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
        elif n == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif n in ("BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG", "BINFLOAT",
                   "BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8",
                   "BINSTRING", "SHORT_BINSTRING"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "MARK":
            stack.append(_MARK)
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(n[-1])
            items = tuple(stack[-k:])
            del stack[-k:]
            stack.append(items)
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "APPENDS":
            items = pop_mark()
            stack[-1].extend(items)
        elif n == "SETITEM":
            v, k = stack.pop(), stack.pop()
            stack[-1][k] = v
        elif n == "SETITEMS":
            items = pop_mark()
            for k, v in zip(items[::2], items[1::2]):
                stack[-1][k] = v
        elif n == "REDUCE":
            args, fn = stack.pop(), stack.pop()
            stack.append(_Sym("call", fn, args))
        elif n == "BUILD":
            state, obj = stack.pop(), stack.pop()
            stack.append(_Sym("built", obj, state))
        elif n == "STOP":
            return stack.pop()
        else:
            raise ValueError(f"pickle opcode {n} is not part of an array container")
    raise ValueError("pickle stream ended without STOP")


def _is_global(node, names) -> bool:
    return isinstance(node, _Sym) and node.kind == "global" and (node.a, node.b) in names


def _array_from_symbolic(node) -> np.ndarray:
    """``ndarray.__reduce__`` as numpy writes it: ``_reconstruct(ndarray, (0,), b'b')`` built with
    ``(version, shape, dtype, is_fortran, raw bytes | list for object arrays)``."""
    recon = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")}
    if not (isinstance(node, _Sym) and node.kind == "built" and isinstance(node.a, _Sym) and node.a.kind == "call"
            and _is_global(node.a.a, recon) and _is_global(node.a.b[0], {("numpy", "ndarray")})):
        raise ValueError("not a pickled ndarray")
    _version, shape, dt, fortran, raw = node.b
    dnode = dt.a if dt.kind == "built" else dt
    if not (dnode.kind == "call" and _is_global(dnode.a, {("numpy", "dtype")}) and isinstance(dnode.b[0], str)):
        raise ValueError("unsupported dtype record")
    code = dnode.b[0]
    if code.startswith("O"):
        return raw                        # object array: its element list (the dict of a 0-d np.save)
    if not (len(code) in (2, 3) and code[-2] in "uifb" and code[-1] in "1248") or fortran:
        raise ValueError(f"unsupported array dtype {code!r}")
    order = dt.b[1] if dt.kind == "built" and dt.b[1] in ("<", ">", "|", "=") else "="
    if isinstance(raw, str):              # protocol-2 pickles carry bytes as latin-1 text
        raw = raw.encode("latin1")
    arr = np.frombuffer(raw, dtype=np.dtype(code).newbyteorder(order))
    if not all(isinstance(d, int) and d >= 0 for d in shape) or int(np.prod(shape, dtype=np.int64)) != arr.size:
        raise ValueError("array payload does not match its shape")
    return arr.reshape(shape).astype(arr.dtype.newbyteorder("="), copy=True)


def read_pickled_event_dict(f) -> dict:
    """The ``{x, y, t, p}`` dict of a pickled 0-d object array, read without unpickling anything."""
    top = _array_from_symbolic(_read_pickle_symbolic(f))
    if isinstance(top, list) and len(top) == 1:
        top = top[0]
    if not isinstance(top, dict):
        raise ValueError("pickled container does not hold a dict")
    return {k: _array_from_symbolic(v) for k, v in top.items() if isinstance(k, str)}


def _validate(ev: dict) -> EventDict:
    missing = [k for k in ("x", "y", "t", "p") if k not in ev]
    if missing:
        raise KeyError(f"event stream lacks keys {missing}")
    out = {k: np.asarray(ev[k]).reshape(-1) for k in ("x", "y", "t", "p")}
    n = {len(v) for v in out.values()}
    if len(n) != 1:
        raise ValueError("x, y, t, p must have equal length")
    return out


def load_event_npy(path: str, trusted: bool = False) -> EventDict:
    """Load an event sample.

    Accepted containers: ``.npz`` with arrays ``x,y,t,p``; a structured ``.npy``
    with those fields; or the pickled-dict ``.npy`` used by EventGPT samples
    (reference: common/common.py:111-112).  Pickled files are read symbolically
    (``read_pickled_event_dict``: no opcode of the file is executed) unless ``trusted``
    (then plain ``np.load`` with ``allow_pickle``).
    """
    if path.endswith(".npz"):
        with np.load(path) as z:
            return _validate({k: z[k] for k in z.files})
    try:
        arr = np.load(path, allow_pickle=False)
    except ValueError:
        if trusted:
            return _validate(np.load(path, allow_pickle=True).item())
        with open(path, "rb") as f:
            version = np.lib.format.read_magic(f)
            if version == (1, 0):
                np.lib.format.read_array_header_1_0(f)
            else:
                np.lib.format.read_array_header_2_0(f)
            return _validate(read_pickled_event_dict(f))
    if arr.dtype.names:
        return _validate({k: arr[k] for k in arr.dtype.names})
    raise ValueError(f"{path}: not an event container")


def save_event_npy(path: str, events: EventDict, pickled: bool = True) -> None:
    """Write an event sample (pickled-dict ``.npy`` by default, else ``.npz``)."""
    ev = _validate(events)
    if pickled and not path.endswith(".npz"):
        np.save(path, np.array(ev, dtype=object), allow_pickle=True)
    else:
        np.savez(path, **ev)


def synthetic_events(n_events: int = 132_268, width: int = 640, height: int = 480,
                     duration_us: int = 49_596, seed: int = 0,
                     compact: bool = True) -> EventDict:
    """Random DVS-like stream with the schema of the EventGPT sample file
    (640×480, ≈50 ms, ``x,y,t:uint16``, ``p:uint8``)."""
    rng = np.random.default_rng(seed)
    # moving blobs give spatial structure instead of uniform noise
    centres = rng.uniform([0, 0], [width, height], size=(8, 2))
    vel = rng.normal(0, 60, size=(8, 2))
    which = rng.integers(0, 8, n_events)
    t = np.sort(rng.integers(0, duration_us, n_events))
    pos = centres[which] + vel[which] * (t[:, None] / max(duration_us, 1)) \
        + rng.normal(0, 25, size=(n_events, 2))
    x = np.clip(pos[:, 0], 0, width - 1).astype(np.int64)
    y = np.clip(pos[:, 1], 0, height - 1).astype(np.int64)
    # guarantee the full sensor extent so the data-dependent canvas is width×height
    x[0], y[0] = width - 1, height - 1
    p = rng.integers(0, 2, n_events)
    if compact:
        return {"x": x.astype(np.uint16), "y": y.astype(np.uint16),
                "t": t.astype(np.uint16 if duration_us < 65536 else np.int64),
                "p": p.astype(np.uint8)}
    return {"x": x, "y": y, "t": t.astype(np.int64), "p": p.astype(np.int64)}


# ------------------------------------------------------------------- slicing
def slice_bounds_by_count(n_events: int, n_slices: int) -> List[Tuple[int, int]]:
    per = n_events // n_slices
    return [(i * per, (i + 1) * per if i < n_slices - 1 else n_events)
            for i in range(n_slices)]


def split_by_count(events: EventDict, n_slices: int) -> List[EventDict]:
    ev = _validate(events)
    return [{k: v[a:b] for k, v in ev.items()}
            for a, b in slice_bounds_by_count(len(ev["t"]), n_slices)]


def split_by_time(events: EventDict, time_interval: int = 50_000) -> List[EventDict]:
    """One dict per occupied ``time_interval``-µs bin, in increasing bin order."""
    ev = _validate(events)
    bins = ev["t"].astype(np.int64) // int(time_interval)
    return [{k: v[bins == b] for k, v in ev.items()} for b in np.unique(bins)]


def check_stream_length(start_time: int, end_time: int, limit_us: int = 100_000) -> None:
    if end_time - start_time >= limit_us:
        raise ValueError("event streams of 100 ms or longer are not supported by "
                         "the 5-slice encoder; split the stream first")


# -------------------------------------------------------------- rasterisation
def canvas_size(x: np.ndarray, y: np.ndarray) -> Tuple[int, int]:
    """Data-dependent canvas ``(H, W) = (y.max()+1, x.max()+1)``."""
    return int(y.max()) + 1, int(x.max()) + 1


def rasterize(x, y, p, size: Optional[Tuple[int, int]] = None,
              conflict: str = "last") -> np.ndarray:
    """Paint one slice to an ``H×W×3`` uint8 image (vectorised, order-exact).

    ``conflict="last"`` reproduces a sequential overwrite: for each pixel the
    event with the highest stream index decides the colour.  Implemented as a
    scatter-max of ``index*2+polarity`` so no Python loop over events is needed.
    """
    x = np.asarray(x).astype(np.int64)
    y = np.asarray(y).astype(np.int64)
    pos = np.asarray(p) != 0
    h, w = size if size is not None else canvas_size(x, y)
    img = np.full((h, w, 3), 255, dtype=np.uint8)
    if len(x) == 0:
        return img
    flat = y * w + x
    if conflict == "last":
        key = np.arange(len(x), dtype=np.int64) * 2 + pos
        winner = np.full(h * w, -1, dtype=np.int64)
        np.maximum.at(winner, flat, key)
        hit = winner >= 0
        red = hit & ((winner & 1) == 1)
        blue = hit & ~red
    elif conflict == "negative":
        has_pos = np.zeros(h * w, dtype=bool)
        has_neg = np.zeros(h * w, dtype=bool)
        has_pos[flat[pos]] = True
        has_neg[flat[~pos]] = True
        blue = has_neg
        red = has_pos & ~has_neg
    else:
        raise ValueError(f"unknown conflict policy {conflict!r}")
    flat_img = img.reshape(-1, 3)
    flat_img[red] = _RED
    flat_img[blue] = _BLUE
    return img


def rasterize_sequential(x, y, p) -> np.ndarray:
    """Literal per-event loop; slow — exists as the test oracle for ``rasterize``."""
    h, w = canvas_size(np.asarray(x), np.asarray(y))
    img = np.full((h, w, 3), 255, dtype=np.uint8)
    for xi, yi, pi in zip(x, y, p):
        img[yi, xi] = _BLUE if pi == 0 else _RED
    return img


def event_frames(events: EventDict, n_frames: int = 5, conflict: str = "last") -> List[np.ndarray]:
    """``n_frames`` rasters from equal-count slices (each on its own canvas)."""
    return [rasterize(s["x"], s["y"], s["p"], conflict=conflict)
            for s in split_by_count(events, n_frames)]


# ---------------------------------------------------------------- voxel grid
def voxel_grid(events: EventDict, bins: int = 5, size: Optional[Tuple[int, int]] = None,
               signed: bool = True, by: str = "time") -> np.ndarray:
    """``[bins, H, W]`` float32 event count volume.

    ``by="time"`` bins by timestamp (uniform over the stream's span),
    ``by="count"`` by stream index.  ``signed`` accumulates +1/−1 by polarity,
    otherwise plain counts.  This is the input format of the reconstruction
    bridge (reference: feasible/feature_alignment/reconstruction.py:107-121).
    """
    ev = _validate(events)
    x = ev["x"].astype(np.int64)
    y = ev["y"].astype(np.int64)
    h, w = size if size is not None else canvas_size(x, y)
    n = len(x)
    vol = np.zeros((bins, h, w), dtype=np.float32)
    if n == 0:
        return vol
    if by == "time":
        t = ev["t"].astype(np.float64)
        span = max(t.max() - t.min(), 1.0)
        b = np.minimum(((t - t.min()) / span * bins).astype(np.int64), bins - 1)
    elif by == "count":
        b = np.minimum(np.arange(n) * bins // n, bins - 1)
    else:
        raise ValueError(by)
    val = np.where(ev["p"] != 0, 1.0, -1.0 if signed else 1.0).astype(np.float32)
    np.add.at(vol, (b, y, x), val)
    return vol


def polarity_histogram(events: EventDict, bins: int = 5,
                       size: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """``[bins, 2, H, W]`` int32 per-time-bin, per-polarity event counts."""
    ev = _validate(events)
    x = ev["x"].astype(np.int64)
    y = ev["y"].astype(np.int64)
    h, w = size if size is not None else canvas_size(x, y)
    hist = np.zeros((bins, 2, h, w), dtype=np.int32)
    if len(x) == 0:
        return hist
    t = ev["t"].astype(np.float64)
    span = max(t.max() - t.min(), 1.0)
    b = np.minimum(((t - t.min()) / span * bins).astype(np.int64), bins - 1)
    np.add.at(hist, (b, (ev["p"] != 0).astype(np.int64), y, x), 1)
    return hist


# ----------------------------------------------------------------- image IO
def load_image(image_file: str):
    """PIL RGB image from a local path or an http(s) URL
    (reference: common/common.py:9-15)."""
    from PIL import Image
    if image_file.startswith(("http://", "https://")):
        import requests
        return Image.open(io.BytesIO(requests.get(image_file).content)).convert("RGB")
    return Image.open(image_file).convert("RGB")
