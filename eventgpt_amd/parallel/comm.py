"""Push all-reduces over peer-mapped buffers: one-shot for the tensor-parallel decode step,
two-shot for the prefill.

A decode step of a tensor-parallel 7B decoder issues 64 all-reduces of ``[B, 4096]`` partial
sums (16 KB at one stream, 512 KB at 32): latency-bound messages for which a ring over the
xGMI mesh uses one of a GPU's seven links per hop.  ``csrc/kernels/comm.hip`` instead has
every rank push its partial into a slot of each peer's receive buffer, flag it, and sum the
slots locally in rank order, fused with the split-K slab sum in front and the residual add
(+ next RMSNorm) behind -- one launch per collective, recorded into the decode hipGraph
like any other kernel.  This module owns the buffers:

* ``OneShotAllReduce`` -- one rank of a real process group: allocates the uncached receive
  buffer and flag words, exchanges ``hipIpcMemHandle``s through the group, maps the peers.
* ``VirtualGroup`` -- W ranks inside ONE process on one GPU (all of them driven by a single
  launch, every block co-resident): how the kernel and its protocol are tested where only one
  GPU is at hand.
* ``TwoShotAllReduce`` / ``VirtualTwoShot`` -- the same pair for the bandwidth-bound messages of a
  sharded prefill (``[rows, hidden]`` bf16): rank r owns slice r, the other ranks push their pieces
  of it into r's staging slots, r sums them in fp32 and pushes the result into every peer's second
  slot (reduce-scatter + all-gather as direct writes over all links at once).

RCCL stays the selectable default (``EVGPT_TP_ALLREDUCE=rccl|oneshot|auto``); ``auto`` runs
``self_test`` against ``dist.all_reduce`` once at start-up and keeps RCCL on any mismatch or
timeout.  Every device-side wait is bounded; ``check()`` raises if one ever gave up.

The reference has no collective at all (its only process group is a world-size-1 NCCL init,
feasible_imu/benchmark_onellm_5stages.py:400-404).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

DEFAULT_SPIN_LIMIT = 1 << 18      # polls (each ~1-2 us) before a wait gives up: a few hundred ms


def _raw():
    from ..ops import hip
    return hip.raw


def plan(m: int, n: int):
    """``(blocks, chunk, blocks_per_row)`` of an ``[m, n]`` message, or ``None`` if it does
    not fit the receive buffers (the caller then takes the library all-reduce)."""
    blocks, chunk, bpr = _raw().comm_plan(int(m), int(n))
    return None if blocks == 0 else (int(blocks), int(chunk), int(bpr))


class _RankBuffers:
    """Receive buffer, flags, epoch counters and error words of one rank on ``device``."""

    def __init__(self, device: torch.device):
        raw = _raw()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.data = raw.comm_alloc(int(raw.comm_data_bytes()), idx)
        self.flag = raw.comm_alloc(int(raw.comm_flag_bytes()), idx)
        self.epoch = torch.zeros(64, dtype=torch.int32, device=device)
        self.err = torch.zeros(4, dtype=torch.int32, device=device)


class _AllReduceBase:
    world: int
    _states: List[torch.Tensor]
    _errs: List[torch.Tensor]

    def check(self) -> None:
        """Raise if a device-side wait ever timed out (reads 16 bytes; call off the hot path)."""
        for r, e in enumerate(self._errs):
            v = e.tolist()
            if v[0]:
                raise RuntimeError(f"one-shot all-reduce: {v[0]} wait(s) timed out on rank slot {r} "
                                   f"(last: epoch {v[1]}, waiting for rank {v[2]}, region {v[3]})")

    def clear_errors(self) -> None:
        for e in self._errs:
            e.zero_()


class OneShotAllReduce(_AllReduceBase):
    """This process's rank of a tensor-parallel group (one process per GPU)."""

    def __init__(self, group, device: torch.device, spin_limit: int = DEFAULT_SPIN_LIMIT):
        raw = _raw()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if not 2 <= self.world <= 8:
            raise ValueError("one-shot all-reduce: 2..8 ranks")
        self.device = device
        self.buf = _RankBuffers(device)
        mine = (raw.comm_ipc_handle(self.buf.data).tolist(), raw.comm_ipc_handle(self.buf.flag).tolist())
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self._opened: List[int] = []
        data_ptrs, flag_ptrs = [], []
        for r, (hd, hf) in enumerate(handles):
            if r == self.rank:
                data_ptrs.append(self.buf.data.data_ptr())
                flag_ptrs.append(self.buf.flag.data_ptr())
                continue
            pd = int(raw.comm_ipc_open(torch.tensor(hd, dtype=torch.uint8), idx))
            pf = int(raw.comm_ipc_open(torch.tensor(hf, dtype=torch.uint8), idx))
            self._opened += [pd, pf]
            data_ptrs.append(pd)
            flag_ptrs.append(pf)
        self.state = raw.comm_make_state(data_ptrs, flag_ptrs, self.buf.epoch, self.buf.err,
                                         self.rank, self.world, int(spin_limit))
        self._states = [self.state]
        self._errs = [self.buf.err]
        # nobody may push before every rank has mapped and zeroed its buffers
        dist.barrier(group=group)

    def close(self) -> None:
        raw = _raw()
        for p in self._opened:
            raw.comm_ipc_close(p)
        self._opened = []

    def reduce(self, partial: torch.Tensor, res: Optional[torch.Tensor] = None,
               x_out: Optional[torch.Tensor] = None, norm_w: Optional[torch.Tensor] = None,
               xn_out: Optional[torch.Tensor] = None, sum_out: Optional[torch.Tensor] = None,
               eps: float = 1e-6) -> bool:
        """All-reduce ``partial [nslab, M, N]`` (fp32) over the group in one launch:
        ``x_out = bf16(sum) + res`` and, for whole-row regions, ``xn_out = RMSNorm(x_out) * norm_w``.
        Returns False without launching if the message does not fit (take RCCL then)."""
        return bool(_raw().comm_allreduce([self.state], [partial], [res], [x_out], norm_w, [xn_out],
                                          [sum_out], float(eps), self.world))

    def self_test(self, iters: int = 3) -> bool:
        """Same answer as ``dist.all_reduce`` on random data, for a one-row and a 32-row
        message, eagerly and from a replayed graph, with no timeout?  Collective: every rank
        of the group calls it; all ranks return the same verdict."""
        ok = True
        g = torch.Generator(device=self.device).manual_seed(1234 + self.rank)
        for m in (1, 32):
            part = torch.randn(2, m, 4096, device=self.device, generator=g)
            want = part.sum(0)
            dist.all_reduce(want, group=self.group)
            got = torch.empty(m, 4096, device=self.device)
            for _ in range(iters):
                got.zero_()
                if not self.reduce(part, sum_out=got):
                    ok = False
                ok = ok and bool(torch.allclose(got, want, rtol=1e-5, atol=1e-4))
        try:
            self.check()
        except RuntimeError:
            ok = False
            self.clear_errors()
        verdict = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(verdict.item()))


class VirtualGroup(_AllReduceBase):
    """``world`` ranks in one process on one GPU; a single launch drives all of them."""

    def __init__(self, world: int, device: torch.device, spin_limit: int = DEFAULT_SPIN_LIMIT):
        raw = _raw()
        if not 1 <= world <= 8:
            raise ValueError("virtual group: 1..8 ranks")
        self.world = world
        self.device = device
        self.bufs = [_RankBuffers(device) for _ in range(world)]
        data_ptrs = [b.data.data_ptr() for b in self.bufs]
        flag_ptrs = [b.flag.data_ptr() for b in self.bufs]
        self._states = [raw.comm_make_state(data_ptrs, flag_ptrs, b.epoch, b.err, r, world, int(spin_limit))
                        for r, b in enumerate(self.bufs)]
        self._errs = [b.err for b in self.bufs]

    def reduce(self, partials: Sequence[torch.Tensor], res: Optional[Sequence] = None,
               x_out: Optional[Sequence] = None, norm_w: Optional[torch.Tensor] = None,
               xn_out: Optional[Sequence] = None, sum_out: Optional[Sequence] = None,
               eps: float = 1e-6, ranks: Optional[Sequence[int]] = None) -> bool:
        """One launch for the ranks in ``ranks`` (default: all).  Driving only some of them
        leaves the others' flags unset: the bounded waits then time out, which is how the
        give-up path is tested without ever hanging a kernel."""
        ranks = list(range(self.world)) if ranks is None else list(ranks)
        none = [None] * len(ranks)
        pick = (lambda seq: none if seq is None else [seq[r] for r in ranks])
        return bool(_raw().comm_allreduce([self._states[r] for r in ranks], [partials[r] for r in ranks],
                                          pick(res), pick(x_out), norm_w, pick(xn_out), pick(sum_out),
                                          float(eps), self.world))


# ----------------------------------------------------------------------- two-shot (large messages)
class _TwoShotBuffers:
    """Scatter / gather staging, flags, epochs and error words of one rank for messages of up to
    ``max_elems`` bf16 elements over ``world`` ranks."""

    def __init__(self, device: torch.device, world: int, max_elems: int):
        raw = _raw()
        piece = int(raw.comm2_piece())
        per_rank = -(-int(max_elems) // world)
        self.pieces_cap = max(1, -(-per_rank // piece))
        stage_bytes = 2 * world * self.pieces_cap * piece * 2
        flag_bytes = -(-(2 * 2 * world * self.pieces_cap * 4) // 16) * 16
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.stage1 = raw.comm_alloc(stage_bytes, idx)
        self.stage2 = raw.comm_alloc(stage_bytes, idx)
        self.flag = raw.comm_alloc(flag_bytes, idx)
        self.epoch = torch.zeros(self.pieces_cap, dtype=torch.int32, device=device)
        self.err = torch.zeros(4, dtype=torch.int32, device=device)


class TwoShotAllReduce(_AllReduceBase):
    """Reduce-scatter + all-gather over peer-mapped staging buffers for the bandwidth-bound
    all-reduces of a sharded prefill (``[S, hidden]`` bf16).  This process's rank of the group."""

    def __init__(self, group, device: torch.device, max_elems: int, spin_limit: int = DEFAULT_SPIN_LIMIT):
        raw = _raw()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if not 2 <= self.world <= 8:
            raise ValueError("two-shot all-reduce: 2..8 ranks")
        self.device = device
        self.buf = _TwoShotBuffers(device, self.world, max_elems)
        mine = tuple(raw.comm_ipc_handle(t).tolist() for t in (self.buf.stage1, self.buf.stage2, self.buf.flag))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self._opened: List[int] = []
        ptrs = ([], [], [])
        for r, hs in enumerate(handles):
            for k, (h, own) in enumerate(zip(hs, (self.buf.stage1, self.buf.stage2, self.buf.flag))):
                if r == self.rank:
                    ptrs[k].append(own.data_ptr())
                else:
                    p = int(raw.comm_ipc_open(torch.tensor(h, dtype=torch.uint8), idx))
                    self._opened.append(p)
                    ptrs[k].append(p)
        self.state = raw.comm2_make_state(ptrs[0], ptrs[1], ptrs[2], self.buf.epoch, self.buf.err,
                                          self.buf.pieces_cap, self.rank, self.world, int(spin_limit))
        self._states = [self.state]
        self._errs = [self.buf.err]
        dist.barrier(group=group)       # nobody pushes before every rank has mapped and zeroed its buffers

    def close(self) -> None:
        raw = _raw()
        for p in self._opened:
            raw.comm_ipc_close(p)
        self._opened = []

    def reduce(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> bool:
        """Sum ``x`` (contiguous bf16, a multiple of 8 elements) over the group into ``out`` (default:
        in place); fp32 accumulation in rank order, the same bits on every rank.  Returns False without
        launching if the message is larger than the buffers were sized for (take RCCL then)."""
        out = x if out is None else out
        return bool(_raw().comm2_allreduce([self.state], [x], [out], self.buf.pieces_cap, self.world))


class VirtualTwoShot(_AllReduceBase):
    """``world`` two-shot ranks in one process on one GPU; a single launch drives all of them.
    Every block of every rank must be resident at once (a rank's block waits for the same block of
    the other ranks), so messages here stay small: world * pieces blocks <= a few hundred."""

    def __init__(self, world: int, device: torch.device, max_elems: int, spin_limit: int = DEFAULT_SPIN_LIMIT):
        raw = _raw()
        if not 2 <= world <= 8:
            raise ValueError("virtual two-shot group: 2..8 ranks")
        self.world, self.device = world, device
        self.bufs = [_TwoShotBuffers(device, world, max_elems) for _ in range(world)]
        s1 = [b.stage1.data_ptr() for b in self.bufs]
        s2 = [b.stage2.data_ptr() for b in self.bufs]
        fl = [b.flag.data_ptr() for b in self.bufs]
        self.pieces_cap = self.bufs[0].pieces_cap
        self._states = [raw.comm2_make_state(s1, s2, fl, b.epoch, b.err, b.pieces_cap, r, world, int(spin_limit))
                        for r, b in enumerate(self.bufs)]
        self._errs = [b.err for b in self.bufs]

    def reduce(self, xs: Sequence[torch.Tensor], outs: Optional[Sequence[torch.Tensor]] = None,
               ranks: Optional[Sequence[int]] = None) -> bool:
        ranks = list(range(self.world)) if ranks is None else list(ranks)
        outs = xs if outs is None else outs
        return bool(_raw().comm2_allreduce([self._states[r] for r in ranks], [xs[r] for r in ranks],
                                           [outs[r] for r in ranks], self.pieces_cap, self.world))


def allreduce_mode() -> str:
    """``rccl`` (default), ``oneshot`` or ``auto`` from ``EVGPT_TP_ALLREDUCE``."""
    m = os.environ.get("EVGPT_TP_ALLREDUCE", "rccl").lower()
    if m not in ("rccl", "oneshot", "auto"):
        raise ValueError(f"EVGPT_TP_ALLREDUCE={m!r}: rccl, oneshot or auto")
    return m
