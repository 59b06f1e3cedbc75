"""DualModelScheduler — drafter and verifier running concurrently.

Two placements:
  * one GPU: each model's work is enqueued from its own host thread onto its own
    HIP stream (default priority), so the drafter's short decode kernels fill the
    CUs the verifier's long prefill leaves idle and vice versa;
  * two GPUs: each model lives on its own device; the only traffic is the hand-off
    of <= 64 token ids (or <= 64 x H bf16 hidden rows) over one xGMI link, a plain
    peer copy.

``parallel_prefill`` measures real overlap (reference: parallel_prefill,
pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:644-715).  ``prefill_hiding``
is the live version of what the reference only *simulated* from separately
measured times (sequential_egpt_vl_prefill :722-853): the drafter prefills and
then keeps decoding until the verifier's prefill completes; the tokens it
produced in that window are free drafts, checked by one batched verify.
"""
from __future__ import annotations

import dataclasses
import threading
import time
from typing import Callable, Dict, Optional

import torch

from .spec_decode import AdapterDraftHead, SDStats, SpecDecodeEngine, _emitted
from .. import ops


def _run_threads(fns):
    """Run callables on their own threads; re-raise the first exception."""
    errs = [None] * len(fns)
    outs = [None] * len(fns)

    def wrap(i):
        try:
            outs[i] = fns[i]()
        except BaseException as e:  # boxed and re-raised on the caller's thread
            errs[i] = e

    ts = [threading.Thread(target=wrap, args=(i,)) for i in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return outs


class DualModelScheduler:
    def __init__(self, drafter_engine, verifier_engine,
                 draft_head: Optional[AdapterDraftHead] = None):
        self.d, self.v = drafter_engine, verifier_engine
        self.head = draft_head
        self.same_device = self.d.device == self.v.device
        self.d_stream = torch.cuda.Stream(device=self.d.device)
        self.v_stream = torch.cuda.Stream(device=self.v.device)
        self.d_lock, self.v_lock = threading.Lock(), threading.Lock()

    def _sync_all(self):
        torch.cuda.synchronize(self.d.device)
        if not self.same_device:
            torch.cuda.synchronize(self.v.device)

    # ------------------------------------------------------------------ prefill
    def _timed_prefill(self, engine, stream, embeds, lock):
        with lock, torch.cuda.device(engine.device), torch.cuda.stream(stream):
            stream.synchronize()
            t0 = time.perf_counter()
            engine.reset()
            engine.prefill(embeds, select="greedy")
            stream.synchronize()      # stream-local: does not serialise the other model
            return (time.perf_counter() - t0) * 1e3

    @torch.no_grad()
    def parallel_prefill(self, d_embeds, v_embeds) -> Dict[str, float]:
        self._sync_all()
        t0 = time.perf_counter()
        d_ms, v_ms = _run_threads([
            lambda: self._timed_prefill(self.d, self.d_stream, d_embeds, self.d_lock),
            lambda: self._timed_prefill(self.v, self.v_stream, v_embeds, self.v_lock)])
        self._sync_all()
        wall = (time.perf_counter() - t0) * 1e3
        seq = d_ms + v_ms
        overlap = max(0.0, seq - wall)
        return {"wall_ms": wall, "egpt_ms": d_ms, "vl_ms": v_ms, "sequential_ms": seq,
                "overlap_ms": overlap, "overlap_ratio": overlap / seq if seq > 0 else 0.0}

    @torch.no_grad()
    def sequential_prefill(self, d_embeds, v_embeds) -> Dict[str, float]:
        self._sync_all()
        d_ms = self._timed_prefill(self.d, self.d_stream, d_embeds, self.d_lock)
        v_ms = self._timed_prefill(self.v, self.v_stream, v_embeds, self.v_lock)
        return {"egpt_ms": d_ms, "vl_ms": v_ms, "sequential_ms": d_ms + v_ms}

    # ----------------------------------------------------------- prefill hiding
    @torch.no_grad()
    def hidden_prefill(self, d_embeds, v_embeds, max_free: int = 30, eos_token_id: int = 2,
                       token_map: Optional[Callable] = None, want_verifier_hidden: bool = False) -> Dict:
        """Drafter prefill + free-running decode in parallel with the verifier's prefill.

        Returns the draft tokens produced inside the window (from ``draft_head``
        applied to the drafter's hidden states when one is configured, else the
        drafter's own tokens, optionally translated by ``token_map`` when the
        vocabularies differ), the timings, and — on request — the verifier hidden
        row that predicts its second token.  The verifier is left prefilled with
        its first token committed.
        """
        stop = threading.Event()
        d, v = self.d, self.v
        for eng in (d, v):      # captures happen here, not inside the threaded window
            with torch.cuda.device(eng.device):
                eng.ensure_graphs(eos_token_id)
        hidden_rows = []
        v_hidden = []

        def run_drafter():
            with self.d_lock, torch.cuda.device(d.device), torch.cuda.stream(self.d_stream):
                t0 = time.perf_counter()
                d.reset()
                h = d.prefill(d_embeds, select="greedy", return_hidden=self.head is not None)
                if self.head is not None:
                    hidden_rows.append(ops.rmsnorm(h[-1:], d.final_norm, d.eps))
                self.d_stream.synchronize()
                t_prefill = (time.perf_counter() - t0) * 1e3
                n = 1
                while n < max_free and not stop.is_set():
                    if self.head is not None:
                        d._layers_step()
                        hidden_rows.append(ops.rmsnorm(d.x[:1].clone(), d.final_norm, d.eps))
                        d._logits_and_commit(greedy=True)
                    else:
                        d.decode_step(True)
                    n += 1
                    if n % 4 == 0:      # bound the run-ahead of the host over the stream
                        self.d_stream.synchronize()
                        if int(d.done[0].item()):
                            break
                self.d_stream.synchronize()
                return {"prefill_ms": t_prefill, "total_ms": (time.perf_counter() - t0) * 1e3,
                        "n_free": min(n, int(d.step[0].item()))}

        def run_verifier():
            try:
                with self.v_lock, torch.cuda.device(v.device), torch.cuda.stream(self.v_stream):
                    self.v_stream.synchronize()
                    t0 = time.perf_counter()
                    v.reset()
                    h = v.prefill(v_embeds, select="greedy", return_hidden=want_verifier_hidden)
                    if want_verifier_hidden:
                        v_hidden.append(ops.rmsnorm(h[-1:], v.final_norm, v.eps))
                    self.v_stream.synchronize()
                    return (time.perf_counter() - t0) * 1e3
            finally:
                stop.set()

        self._sync_all()
        t0 = time.perf_counter()
        d_info, v_ms = _run_threads([run_drafter, run_verifier])
        self._sync_all()
        wall_prefill = (time.perf_counter() - t0) * 1e3
        n_free = d_info["n_free"]
        with torch.cuda.device(v.device):
            if self.head is not None:
                draft = self.head(torch.cat(hidden_rows)[:n_free].to(v.device))
            else:
                draft = _emitted(d)[:n_free].to(v.device)
                if token_map is not None:
                    draft = token_map(draft)
        return {"draft": draft.to(torch.int32), "n_free": n_free, "egpt_prefill_ms": d_info["prefill_ms"],
                "egpt_total_ms": d_info["total_ms"], "vl_prefill_ms": v_ms, "prefill_wall_ms": wall_prefill,
                "verifier_hidden": v_hidden[0] if v_hidden else None}

    @torch.no_grad()
    def prefill_hiding(self, d_embeds, v_embeds, max_new_tokens: int = 50, max_free: int = 30,
                       eos_token_id: int = 2, token_map: Optional[Callable] = None) -> Dict:
        """``hidden_prefill`` + one batched verify of the free tokens + autoregressive
        continuation on the verifier."""
        pre = self.hidden_prefill(d_embeds, v_embeds, max_free, eos_token_id, token_map)
        with torch.cuda.device(self.v.device):
            sd = SpecDecodeEngine(self.v, self.head)
            toks, stats = sd.verify_free_tokens(pre["draft"], max_new_tokens, eos_token_id, drop_first=True)
        return {"tokens": toks, "stats": stats.as_dict(), "n_free": pre["n_free"],
                "egpt_prefill_ms": pre["egpt_prefill_ms"], "vl_prefill_ms": pre["vl_prefill_ms"],
                "prefill_wall_ms": pre["prefill_wall_ms"], "decode_ms": stats.decode_ms,
                "total_ms": pre["prefill_wall_ms"] + stats.decode_ms}

    # ----------------------------------------------------------------- baseline
    @torch.no_grad()
    def verifier_baseline(self, v_embeds, max_new_tokens: int = 50, eos_token_id: int = 2) -> Dict:
        """Plain autoregressive decode on the verifier (the number SD has to beat)."""
        v = self.v
        with torch.cuda.device(v.device):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            v.reset()
            v.prefill(v_embeds, select="greedy")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            toks = v.generate_continue(max_new_tokens, eos_token_id)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        return {"tokens": toks, "prefill_ms": (t1 - t0) * 1e3, "decode_ms": (t2 - t1) * 1e3,
                "total_ms": (t2 - t0) * 1e3}
