"""DSEC-style event recordings: file access, index/time-window slicing, directory layout.

A recording is ``{events/{x,y,t,p}, ms_to_idx, t_offset}``: ``t`` is microseconds
relative to ``t_offset`` and sorted, ``ms_to_idx[k]`` is the index of the first event
with ``t >= k*1000`` (reference: dataset/io.py:38-95, dataset/directory.py:11-54).
Two containers are understood: HDF5 (needs ``h5py``, which is optional) and an
``.npz`` with the same member names (``x, y, t, p, ms_to_idx, t_offset``), which is
what this package writes.  Slicing by time uses ``ms_to_idx`` when present and a
binary search on the sorted timestamps otherwise — never a full-array mask.
"""
from __future__ import annotations

import filecmp
import json
import os
from functools import cached_property
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

EventDict = Dict[str, np.ndarray]


def build_ms_to_idx(t: np.ndarray) -> np.ndarray:
    """``out[k]`` = index of the first event with ``t >= k * 1000`` (one extra entry at the end)."""
    if t.size == 0:
        return np.zeros(1, dtype=np.int64)
    n_ms = int(t[-1]) // 1000 + 2
    return np.searchsorted(t, np.arange(n_ms, dtype=np.int64) * 1000, side="left").astype(np.int64)


def save_recording(path: str, events: EventDict, t_offset: int = 0) -> str:
    """Write an ``.npz`` recording; ``events['t']`` must be sorted, relative to ``t_offset``."""
    t = np.asarray(events["t"], dtype=np.int64)
    if t.size > 1 and np.any(np.diff(t) < 0):
        raise ValueError("event timestamps must be sorted")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, x=np.asarray(events["x"], dtype=np.uint16), y=np.asarray(events["y"], dtype=np.uint16), t=t,
             p=np.asarray(events["p"], dtype=np.uint8), ms_to_idx=build_ms_to_idx(t),
             t_offset=np.int64(t_offset))
    return path


class EventFile:
    """Random access to one recording (context manager)."""

    def __init__(self, path):
        self.path = str(path)
        self._h5 = None
        if self.path.endswith((".h5", ".hdf5")):
            try:
                import h5py
            except ImportError as e:
                raise ImportError("reading .h5 recordings needs h5py; convert to .npz with the same members") from e
            self._h5 = h5py.File(self.path, "r")
            ev = self._h5["events"]
            self.x, self.y, self.t, self.p = ev["x"], ev["y"], ev["t"], ev["p"]
            self.t_offset = int(self._h5["t_offset"][()]) if "t_offset" in self._h5 else 0
            self.ms_to_idx = np.asarray(self._h5["ms_to_idx"], dtype=np.int64) if "ms_to_idx" in self._h5 else None
        else:
            z = np.load(self.path, mmap_mode="r") if self.path.endswith(".npy") else np.load(self.path)
            self.x, self.y, self.t, self.p = z["x"], z["y"], z["t"], z["p"]
            self.t_offset = int(z["t_offset"]) if "t_offset" in z else 0
            self.ms_to_idx = np.asarray(z["ms_to_idx"], dtype=np.int64) if "ms_to_idx" in z else None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self) -> None:
        if self._h5 is not None:
            self._h5.close()
            self._h5 = None

    def __len__(self) -> int:
        return int(self.t.shape[0])

    @property
    def time_range(self):
        """Absolute (offset-applied) first and last timestamps."""
        return int(self.t[0]) + self.t_offset, int(self.t[-1]) + self.t_offset

    def by_index(self, start: int, end: int) -> EventDict:
        return {"p": np.asarray(self.p[start:end]), "t": np.asarray(self.t[start:end]).astype(np.int64) + self.t_offset,
                "x": np.asarray(self.x[start:end]), "y": np.asarray(self.y[start:end])}

    def index_of_time(self, t_abs_us: int) -> int:
        """Index of the first event at or after absolute time ``t_abs_us``."""
        rel = int(t_abs_us) - self.t_offset
        if rel <= 0:
            return 0
        if self.ms_to_idx is not None:
            ms = rel // 1000
            if ms + 1 < len(self.ms_to_idx):
                lo, hi = int(self.ms_to_idx[ms]), int(self.ms_to_idx[ms + 1])
                return lo + int(np.searchsorted(np.asarray(self.t[lo:hi]), rel, side="left"))
            return len(self)
        return int(np.searchsorted(np.asarray(self.t), rel, side="left"))

    def by_timewindow(self, t_min_us: int, t_max_us: int, exact: bool = True) -> EventDict:
        """Events with ``t_min <= t < t_max`` (absolute µs).  ``exact=False`` cuts at the
        millisecond boundaries of ``ms_to_idx`` like the DSEC helper does."""
        if not exact and self.ms_to_idx is not None:
            a = max((int(t_min_us) - self.t_offset) // 1000, 0)
            b = max((int(t_max_us) - self.t_offset) // 1000, 0)
            last = len(self.ms_to_idx) - 1
            return self.by_index(int(self.ms_to_idx[min(a, last)]), int(self.ms_to_idx[min(b, last)]))
        return self.by_index(self.index_of_time(t_min_us), self.index_of_time(t_max_us))


def get_num_events(path) -> int:
    with EventFile(path) as f:
        return len(f)


def extract_from_h5_by_index(path, ev_start_idx: int, ev_end_idx: int) -> EventDict:
    with EventFile(path) as f:
        return f.by_index(ev_start_idx, ev_end_idx)


def extract_from_h5_by_timewindow(path, t_min_us: int, t_max_us: int) -> EventDict:
    with EventFile(path) as f:
        return f.by_timewindow(t_min_us, t_max_us, exact=False)


def h5_file_to_dict(path) -> Dict:
    import h5py
    with h5py.File(str(path), "r") as fh:
        return {k: fh[k][()] for k in fh.keys()}


def yaml_file_to_dict(path) -> Dict:
    import yaml
    with open(path) as fh:
        return yaml.load(fh, Loader=yaml.SafeLoader)


def compare_dirs(dir1, dir2) -> bool:
    """True when the two trees hold the same files with the same bytes."""
    cmp = filecmp.dircmp(str(dir1), str(dir2))
    if cmp.left_only or cmp.right_only or cmp.funny_files:
        return False
    _, mismatch, errors = filecmp.cmpfiles(str(dir1), str(dir2), cmp.common_files, shallow=False)
    if mismatch or errors:
        return False
    return all(compare_dirs(Path(dir1) / d, Path(dir2) / d) for d in cmp.common_dirs)


# --------------------------------------------------------------- directory layout
class BaseDirectory:
    def __init__(self, root):
        self.root = Path(root)


class ImageDirectory(BaseDirectory):
    @cached_property
    def timestamps(self) -> np.ndarray:
        return np.genfromtxt(self.root / "timestamps.txt", dtype="int64")

    @cached_property
    def image_files_rectified(self) -> List[Path]:
        return sorted((self.root / "left" / "rectified").glob("*.png"))

    @cached_property
    def image_files_distorted(self) -> List[Path]:
        return sorted((self.root / "left" / "distorted").glob("*.png"))


class EventDirectory(BaseDirectory):
    @cached_property
    def event_file(self) -> Path:
        for name in ("events.h5", "events.npz"):
            p = self.root / "left" / name
            if p.exists():
                return p
        return self.root / "left" / "events.h5"


class TracksDirectory(BaseDirectory):
    @cached_property
    def tracks(self) -> np.ndarray:
        return np.load(self.root / "left" / "tracks.npy", allow_pickle=False)


class LabelDirectory(BaseDirectory):
    @cached_property
    def labels(self):
        with open(self.root / "QADataset.json") as f:
            return json.load(f)


class DSECDirectory:
    def __init__(self, root):
        self.root = Path(root)
        self.images = ImageDirectory(self.root / "images")
        self.events = EventDirectory(self.root / "events")
        self.tracks = TracksDirectory(self.root / "object_detections")
        self.labels = LabelDirectory(self.root)
