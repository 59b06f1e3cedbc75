"""Stage timers, forward-hook timers and memory accounting for benchmarks.

The reference brackets every stage with ``torch.cuda.synchronize()`` +
``time.time()`` and uses module forward hooks to split a monolithic forward
into vision / prefill / decode (pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:59-93,
feasible/benchmark_inference/benchmark_inference_5stages.py:485-559, 843-852,
feasible/benchmark_inference/benchmark_inference_3stages.py:63-75). Here stages
are HIP-event spans on the current stream, resolved once at the end of a sample,
so timing does not serialise the device between stages; host-only stages (file
load, CPU preprocessing) use the host clock.
"""
from __future__ import annotations

import time
from collections import OrderedDict
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch


def _is_gpu(device) -> bool:
    return device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available()


class StageTimer:
    """``with timer.stage("vision"): ...`` → ``timer.result()`` in milliseconds."""

    def __init__(self, device=None):
        self.gpu = _is_gpu(device)
        self._pending: List[tuple] = []
        self._ms: "OrderedDict[str, float]" = OrderedDict()

    @contextmanager
    def stage(self, name: str, host: bool = False):
        if self.gpu and not host:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            try:
                yield
            finally:
                b.record()
                self._pending.append((name, a, b))
        else:
            if self.gpu:
                torch.cuda.current_stream().synchronize()
            t0 = time.perf_counter()
            try:
                yield
            finally:
                if self.gpu:
                    torch.cuda.current_stream().synchronize()
                self._ms[name] = self._ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3

    def add(self, name: str, ms: float) -> None:
        self._ms[name] = self._ms.get(name, 0.0) + ms

    def result(self) -> Dict[str, float]:
        for name, a, b in self._pending:
            b.synchronize()
            self._ms[name] = self._ms.get(name, 0.0) + a.elapsed_time(b)
        self._pending = []
        return dict(self._ms)

    def reset(self) -> None:
        self._pending, self._ms = [], OrderedDict()


class _HookTimer:
    def __init__(self):
        self.hooks = []

    def _span(self):
        if torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    @staticmethod
    def _elapsed(a, b) -> float:
        if isinstance(a, float):
            return b - a
        b.synchronize()
        return a.elapsed_time(b) / 1e3

    def unregister(self) -> None:
        for h in self.hooks:
            h.remove()
        self.hooks = []


class VisionTimingHooks(_HookTimer):
    """Time the vision tower inside a monolithic forward; ``vision_time`` is in seconds."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self._a = self._b = None

    def register(self) -> bool:
        tower = None
        for getter in ("get_visual_tower", "get_vision_tower"):
            if hasattr(self.model, getter):
                tower = getattr(self.model, getter)()
                break
        if tower is None:
            return False
        # the encode path calls the wrapped transformer directly, so hook that module
        inner = getattr(tower, "visual_tower", None)
        if isinstance(inner, torch.nn.Module):
            tower = inner
        self.hooks.append(tower.register_forward_pre_hook(lambda m, i: setattr(self, "_a", self._span())))
        self.hooks.append(tower.register_forward_hook(lambda m, i, o: setattr(self, "_b", self._span())))
        return True

    @property
    def vision_time(self) -> float:
        if self._a is None or self._b is None:
            return 0.0
        return self._elapsed(self._a, self._b)

    def reset(self) -> None:
        self._a = self._b = None


class PrefillDecodeHooks(_HookTimer):
    """Split a hooked decoder's forwards into the first (prefill) and the rest (decode)."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.reset()

    def _target(self):
        m = self.model
        for name in ("language_model", "model"):
            if hasattr(m, name):
                m = getattr(m, name)
                break
        return getattr(m, "model", m)

    def _pre(self, module, inputs):
        if not self.first_forward_done:
            self._p0 = self._span()
        elif self._d0 is None:
            self._d0 = self._span()

    def _post(self, module, inputs, output):
        if not self.first_forward_done:
            self._p1 = self._span()
            self.first_forward_done = True
        self._d1 = self._span()
        self.forwards += 1

    def register_hooks(self) -> bool:
        try:
            t = self._target()
            self.hooks = [t.register_forward_pre_hook(self._pre), t.register_forward_hook(self._post)]
            return True
        except Exception:
            return False

    unregister_hooks = _HookTimer.unregister

    def finalize_decode_time(self) -> float:
        if self._d0 is not None and self._d1 is not None:
            self.decode_time = self._elapsed(self._d0, self._d1)
        if self._p0 is not None and self._p1 is not None:
            self.prefill_time = self._elapsed(self._p0, self._p1)
        return self.decode_time

    def reset(self) -> None:
        self.prefill_time = self.decode_time = 0.0
        self._p0 = self._p1 = self._d0 = self._d1 = None
        self.first_forward_done = False
        self.forwards = 0


# ------------------------------------------------------------------------ memory
def gpu_memory_mb(device=None) -> Dict[str, float]:
    if not torch.cuda.is_available():
        return {"allocated_mb": 0.0, "reserved_mb": 0.0, "peak_mb": 0.0}
    mb = 1024.0 ** 2
    return {"allocated_mb": torch.cuda.memory_allocated(device) / mb,
            "reserved_mb": torch.cuda.memory_reserved(device) / mb,
            "peak_mb": torch.cuda.max_memory_allocated(device) / mb}


def get_gpu_memory_mb(device=None) -> float:
    return gpu_memory_mb(device)["allocated_mb"]


def get_gpu_memory_peak_mb(device=None) -> float:
    return gpu_memory_mb(device)["peak_mb"]


def estimate_kv_cache_mb(past_key_values, length: Optional[int] = None) -> float:
    """Bytes held by a tuple-of-(k, v) cache; ``length`` prices only the first ``length``
    positions of a preallocated ``[B, H, L, D]`` cache."""
    if past_key_values is None:
        return 0.0
    total = 0
    for k, v in past_key_values:
        for t in (k, v):
            n = t.numel()
            if length is not None and t.dim() == 4 and t.shape[2] > 0:
                n = n // t.shape[2] * min(length, t.shape[2])
            total += n * t.element_size()
    return total / 1024.0 ** 2


def model_memory_mb(model: torch.nn.Module) -> float:
    return sum(p.numel() * p.element_size() for p in model.parameters()) / 1024.0 ** 2
