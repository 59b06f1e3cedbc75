"""Timing utilities: single-shot, averaging and multi-step profilers.

API surface of the reference's profiling helper
(feasible/mllm_profiling_2025/profiler.py:10-385: ``Profiler``,
``AveragingProfiler``, ``MultiStepProfiler``/``ProfileStep``, the
``profile_function[_avg]`` decorators, ``time_block``, ``profile_step[_avg]`` and
the colour helpers). The clock is different: a GPU launch returns before the
kernels run, so wall-clock brackets around asynchronous work measure launch
cost. Every profiler here takes ``device=``: with a GPU device the span is
bracketed by HIP events on the current stream and read after an event
synchronise (no device-wide sync, so other streams keep running); without one
it is ``time.perf_counter``.
"""
from __future__ import annotations

import functools
import statistics
import sys
import time
from contextlib import contextmanager
from typing import Any, Callable, Dict, List, Optional

import torch

ENABLE_COLORS = sys.stdout.isatty()


class Colors:
    RESET, BOLD, DIM = "\033[0m", "\033[1m", "\033[2m"
    BLACK, RED, GREEN, YELLOW, BLUE, MAGENTA, CYAN, WHITE = (f"\033[{c}m" for c in range(30, 38))
    (BRIGHT_BLACK, BRIGHT_RED, BRIGHT_GREEN, BRIGHT_YELLOW, BRIGHT_BLUE, BRIGHT_MAGENTA, BRIGHT_CYAN,
     BRIGHT_WHITE) = (f"\033[{c}m" for c in range(90, 98))
    BG_BLACK, BG_RED, BG_GREEN, BG_YELLOW, BG_BLUE, BG_MAGENTA, BG_CYAN, BG_WHITE = (
        f"\033[{c}m" for c in range(40, 48))


def set_color_enabled(enabled: bool) -> None:
    global ENABLE_COLORS
    ENABLE_COLORS = bool(enabled)


def is_color_enabled() -> bool:
    return ENABLE_COLORS


def colorize_text(text: str, color: str = Colors.WHITE) -> str:
    return f"{color}{text}{Colors.RESET}" if ENABLE_COLORS else text


def colorize_profiling_tag(tag: str = "[PROFILING]") -> str:
    return f"{Colors.BOLD}{Colors.BRIGHT_CYAN}{tag}{Colors.RESET}" if ENABLE_COLORS else tag


_DURATION_BANDS = ((1e-3, Colors.BRIGHT_GREEN), (0.1, Colors.GREEN), (1.0, Colors.YELLOW),
                   (5.0, Colors.BRIGHT_YELLOW), (float("inf"), Colors.RED))


def colorize_duration(duration: float) -> str:
    text = f"{duration:.4f} seconds ({duration * 1000:.2f} ms)"
    return colorize_text(text, next(c for lim, c in _DURATION_BANDS if duration < lim))


def _say(*parts: str) -> None:
    print(colorize_profiling_tag(), *parts)


class _Span:
    """One timed span on the host clock or on a device stream."""

    def __init__(self, device=None):
        dev = torch.device(device) if device is not None else None
        self.gpu = dev is not None and dev.type == "cuda"
        self._t0 = self._t1 = None

    def begin(self) -> None:
        if self.gpu:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t1 = torch.cuda.Event(enable_timing=True)
            self._t0.record()
        else:
            self._t0 = time.perf_counter()

    def end(self) -> None:
        if self.gpu:
            self._t1.record()
        else:
            self._t1 = time.perf_counter()

    def seconds(self) -> Optional[float]:
        if self._t0 is None or self._t1 is None:
            return None
        if self.gpu:
            self._t1.synchronize()
            return self._t0.elapsed_time(self._t1) / 1e3
        return self._t1 - self._t0


class Profiler:
    """Context manager / start-stop timer that prints its result."""

    def __init__(self, name: str = "Operation", device=None, verbose: bool = True):
        self.name, self.verbose = name, verbose
        self._span = _Span(device)
        self.duration: Optional[float] = None

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()
        if self.verbose:
            self.print_result()

    def start(self) -> None:
        if self.verbose:
            _say("Starting", colorize_text(self.name, Colors.BRIGHT_WHITE))
        self._span.begin()

    def stop(self) -> Optional[float]:
        self._span.end()
        self.duration = self._span.seconds()
        return self.duration

    def print_result(self) -> None:
        if self.duration is not None:
            _say(colorize_text(self.name, Colors.BRIGHT_WHITE), "completed in", colorize_duration(self.duration))

    def get_duration(self) -> Optional[float]:
        return self.duration

    def get_duration_ms(self) -> Optional[float]:
        return None if self.duration is None else self.duration * 1e3


class AveragingProfiler:
    """Run a callable ``num_runs`` times (after ``warmup`` untimed runs) and keep statistics."""

    def __init__(self, name: str = "Operation", num_runs: int = 10, device=None, warmup: int = 0,
                 verbose: bool = True):
        self.name, self.num_runs, self.device, self.warmup, self.verbose = name, num_runs, device, warmup, verbose
        self.durations: List[float] = []

    def profile_function(self, func: Callable, *args, **kwargs) -> Any:
        result = None
        for _ in range(self.warmup):
            result = func(*args, **kwargs)
        spans = []
        for _ in range(self.num_runs):
            s = _Span(self.device)
            s.begin()
            result = func(*args, **kwargs)
            s.end()
            spans.append(s)
        self.durations = [s.seconds() for s in spans]
        if self.verbose:
            self.print_statistics()
        return result

    def profile_callable(self, callable_obj: Callable) -> Any:
        return self.profile_function(callable_obj)

    def statistics(self) -> Dict[str, float]:
        d = self.durations
        if not d:
            return {}
        return {"mean": statistics.mean(d), "min": min(d), "max": max(d), "median": statistics.median(d),
                "std": statistics.stdev(d) if len(d) > 1 else 0.0, "runs": len(d)}

    def print_statistics(self) -> None:
        st = self.statistics()
        if not st:
            _say(colorize_text("No timing data available", Colors.RED))
            return
        _say(colorize_text(f"=== {self.name} ({st['runs']} runs) ===", Colors.BRIGHT_MAGENTA))
        for label, key in (("Average:", "mean"), ("Median: ", "median"), ("Minimum:", "min"), ("Maximum:", "max")):
            _say(colorize_text(label, Colors.BRIGHT_WHITE), colorize_duration(st[key]))
        _say(colorize_text("Std Dev:", Colors.BRIGHT_WHITE), f"{st['std'] * 1e3:.3f} ms")

    def get_average_duration(self) -> Optional[float]:
        return statistics.mean(self.durations) if self.durations else None

    def get_average_duration_ms(self) -> Optional[float]:
        avg = self.get_average_duration()
        return None if avg is None else avg * 1e3

    def get_all_durations(self) -> List[float]:
        return list(self.durations)


def profile_function(name: Optional[str] = None, device=None):
    def deco(func):
        @functools.wraps(func)
        def wrapper(*a, **kw):
            with Profiler(name or func.__name__, device):
                return func(*a, **kw)
        return wrapper
    return deco


def profile_function_avg(name: Optional[str] = None, num_runs: int = 10, device=None):
    def deco(func):
        @functools.wraps(func)
        def wrapper(*a, **kw):
            return AveragingProfiler(name or func.__name__, num_runs, device).profile_function(func, *a, **kw)
        return wrapper
    return deco


@contextmanager
def time_block(name: str, device=None):
    with Profiler(name, device) as p:
        yield p


def profile_step(step_name: str, func: Callable, *args, **kwargs) -> Any:
    with Profiler(step_name, kwargs.pop("device", None)):
        return func(*args, **kwargs)


def profile_step_avg(step_name: str, func: Callable, num_runs: int = 10, *args, **kwargs) -> Any:
    return AveragingProfiler(step_name, num_runs, kwargs.pop("device", None)).profile_function(func, *args, **kwargs)


class MultiStepProfiler:
    """Named steps of one process; spans are resolved together at ``end_process``
    so timing a step never forces a synchronisation between steps."""

    def __init__(self, process_name: str = "Process", device=None, verbose: bool = True):
        self.process_name, self.device, self.verbose = process_name, device, verbose
        self._spans: List[tuple] = []
        self._total = _Span(device)
        self.steps: List[tuple] = []
        self.total: Optional[float] = None

    def start_process(self) -> None:
        self._spans, self.steps = [], []
        self._total.begin()

    def step(self, step_name: str) -> "ProfileStep":
        return ProfileStep(self, step_name)

    def add_step_result(self, step_name: str, duration: float) -> None:
        self.steps.append((step_name, duration))

    def end_process(self) -> Dict[str, float]:
        self._total.end()
        self.total = self._total.seconds()
        for name, span in self._spans:
            self.steps.append((name, span.seconds()))
        self._spans = []
        if self.verbose:
            _say(colorize_text(f"=== {self.process_name} ===", Colors.BRIGHT_MAGENTA))
            for name, d in self.steps:
                share = 100.0 * d / self.total if self.total else 0.0
                _say(f"{name:32s}", colorize_duration(d), f"{share:5.1f}%")
            _say(colorize_text("Total:", Colors.BRIGHT_WHITE), colorize_duration(self.total))
        return dict(self.steps)


class ProfileStep:
    def __init__(self, parent: MultiStepProfiler, step_name: str):
        self.parent, self.step_name = parent, step_name
        self._span = _Span(parent.device)

    def __enter__(self):
        self._span.begin()
        return self

    def __exit__(self, *exc):
        self._span.end()
        self.parent._spans.append((self.step_name, self._span))
