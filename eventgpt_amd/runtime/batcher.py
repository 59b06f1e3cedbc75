"""Continuous batching over a multi-slot engine: requests join and leave a running
decode batch one slot at a time.

The engine (``parallel.tp.TensorParallelEngine`` with ``max_batch`` slots) keeps every
sequence's position, step counter, EOS flag and token ring in device memory and advances
all slots with ONE captured decode graph per step.  The scheduler only edits the per-slot
``active`` mask that the commit kernel reads:

  * waiting requests are admitted into free slots, up to ``max_admit_per_step`` at a time: their
    prompts are prefilled as ONE pack (the projections run over all their rows together, RoPE /
    cache append / attention per slot) and their first tokens are committed with every other slot
    masked out;
  * every step replays the graph; finished slots (EOS or token budget) are parked, their
    tokens handed to the request, and the slot is free for the next admission;
  * the host reads ``step``/``done``/new tokens once every ``poll_every`` steps, so the
    steady state has no per-token synchronisation.

The reference has no counterpart (its loops serve one request at a time); this is the
serving shape the 32-stream throughput numbers of ``bench.py --batch`` correspond to.
"""
from __future__ import annotations

import collections
import dataclasses
import time
from typing import Callable, Deque, Dict, List, Optional

import torch


@dataclasses.dataclass
class Request:
    rid: int
    embeds: torch.Tensor
    max_new_tokens: int
    on_finish: Optional[Callable[["Request"], None]] = None
    tokens: List[int] = dataclasses.field(default_factory=list)
    slot: int = -1
    t_submit: float = 0.0
    t_first: float = 0.0
    t_done: float = 0.0
    finished: bool = False

    @property
    def latency_s(self) -> float:
        return self.t_done - self.t_submit


class ContinuousBatcher:
    def __init__(self, engine, eos_token_id: int = 2, poll_every: int = 8, max_admit_per_step: int = 8):
        if not hasattr(engine, "admit"):
            raise TypeError("ContinuousBatcher needs a multi-slot engine (TensorParallelEngine)")
        self.e = engine
        self.e.eos = int(eos_token_id)
        self.e.enable_slot_masks()
        self.e.reset()
        self.e.active.zero_()
        self.poll_every = max(1, poll_every)
        self.max_admit = max(1, max_admit_per_step)
        self.waiting: Deque[Request] = collections.deque()
        self.running: Dict[int, Request] = {}          # slot -> request
        self.free: List[int] = list(range(engine.B))
        self.finished: List[Request] = []
        self._next_id = 0
        self.steps = 0

    # ---------------------------------------------------------------- interface
    def submit(self, embeds: torch.Tensor, max_new_tokens: int = 64,
               on_finish: Optional[Callable[[Request], None]] = None) -> Request:
        if max_new_tokens > self.e.max_out:
            raise ValueError(f"max_new_tokens {max_new_tokens} > engine ring {self.e.max_out}")
        if embeds.shape[0] + max_new_tokens > self.e.l_max:
            raise ValueError("prompt + budget exceed the KV cache of a slot")
        r = Request(self._next_id, embeds, max_new_tokens, on_finish, t_submit=time.perf_counter())
        self._next_id += 1
        self.waiting.append(r)
        return r

    @property
    def idle(self) -> bool:
        return not self.waiting and not self.running

    # ------------------------------------------------------------------ the loop
    def _admit(self) -> None:
        n = min(len(self.free), len(self.waiting), self.max_admit)
        if n == 0:
            return
        reqs = [self.waiting.popleft() for _ in range(n)]
        slots = [self.free.pop() for _ in range(n)]
        self.e.admit(slots, [r.embeds for r in reqs])
        for r, s in zip(reqs, slots):
            r.slot = s
            self.running[s] = r

    def _collect(self) -> None:
        if not self.running:
            return
        e = self.e
        step = e.step.tolist()                 # one small D2H copy each
        done = e.done.tolist()
        e._poll_hook()                         # (tensor-parallel engines: did a collective's wait give up?)
        now = time.perf_counter()
        retire = []
        for s, r in self.running.items():
            n = min(step[s], r.max_new_tokens)
            if n > len(r.tokens):
                new = e.out_tokens[s, len(r.tokens):n].tolist()
                if not r.tokens:
                    r.t_first = now
                r.tokens.extend(new)
            hit_eos = e.eos >= 0 and e.eos in r.tokens
            if hit_eos:
                r.tokens = r.tokens[: r.tokens.index(e.eos) + 1]
            if hit_eos or done[s] or len(r.tokens) >= r.max_new_tokens:
                retire.append(s)
        if retire:
            e.park(retire)
            for s in retire:
                r = self.running.pop(s)
                r.finished, r.t_done = True, now
                self.free.append(s)
                self.finished.append(r)
                if r.on_finish is not None:
                    r.on_finish(r)

    def step(self) -> None:
        """Admit what fits, advance every running sequence by one token."""
        self._admit()
        if self.running:
            self.e.decode_step(True)
            self.steps += 1
            # poll on the cadence, and right away when a request may have hit its budget
            if self.steps % self.poll_every == 0 or any(
                    len(r.tokens) + self.poll_every >= r.max_new_tokens for r in self.running.values()):
                self._collect()

    def run_until_idle(self, max_steps: int = 1_000_000) -> List[Request]:
        for _ in range(max_steps):
            if self.idle:
                break
            self.step()
        self._collect()
        return self.finished

    def stats(self) -> Dict[str, float]:
        fin = self.finished
        toks = sum(len(r.tokens) for r in fin)
        if not fin:
            return {"requests": 0, "tokens": 0}
        t0, t1 = min(r.t_submit for r in fin), max(r.t_done for r in fin)
        return {"requests": len(fin), "tokens": toks, "wall_s": t1 - t0,
                "tokens_per_s": toks / (t1 - t0) if t1 > t0 else 0.0,
                "mean_latency_s": sum(r.latency_s for r in fin) / len(fin),
                "mean_ttft_s": sum(r.t_first - r.t_submit for r in fin) / len(fin), "decode_steps": self.steps}
