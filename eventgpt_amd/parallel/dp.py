"""Data-parallel training with sharded optimizer state (reduce-scatter / all-gather).

Every rank holds the full (small) model, runs forward/backward on its own
micro-batch, and then — instead of all-reducing gradients and stepping a
replicated optimizer — does, per *bucket* of parameters:

    grads  --reduce-scatter(avg)-->  my 1/W shard of the averaged gradient
    AdamW on my shard (fp32 master weights + moments live only for that shard)
    params <--all-gather--           every rank's updated shard

Buckets are flat, padded to a multiple of the world size, and sized for xGMI:
the fabric is point-to-point (7 links of ~153 GB/s per GPU), where a ring
collective is bound by one link, so few large messages win over many small
ones — 64 MiB buckets put an entire L1..L3 adapter in one collective and
an L4..L5F adapter (100-170 M parameters) in 3-6.  Gradients are written by autograd
directly into the flat bucket (``param.grad`` is a view), and a bucket's
reduce-scatter is launched from a post-accumulate hook as soon as its last
gradient lands, on a side stream, so communication overlaps the rest of backward.
Gradient clipping uses the global norm assembled from the shards with one scalar
all-reduce; norm and clip factor stay on the device (the fused AdamW kernel reads the
factor), and each bucket's all-gather runs on the communication stream under the next
bucket's AdamW.  The reference trains single-process (train_hidden_adapter.py:320-359).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], world: int, rank: int):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.padded = (self.numel + world - 1) // world * world
        self.shard_len = self.padded // world
        p0 = params[0]
        self.flat_param = torch.zeros(self.padded, dtype=p0.dtype, device=p0.device)
        self.flat_grad = torch.zeros(self.padded, dtype=p0.dtype, device=p0.device)
        self.grad_views = []
        off = 0
        for p in params:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            p.grad = self.flat_grad[off:off + n].view_as(p)
            self.grad_views.append(p.grad)
            off += n
        lo = rank * self.shard_len
        self.shard_slice = slice(lo, lo + self.shard_len)
        # fp32 master copy and AdamW moments exist only for this rank's shard
        self.master = self.flat_param[self.shard_slice].float().clone()
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        # the averaged gradient of this rank's shard; with one rank that is the gradient itself
        self.grad_shard = (self.flat_grad[self.shard_slice] if world == 1
                           else torch.zeros(self.shard_len, dtype=p0.dtype, device=p0.device))
        # staging for the refreshed shard when it has to be all-gathered (world > 1)
        self.new_shard = torch.empty_like(self.grad_shard) if world > 1 else None
        self.pending = 0
        self.work = None
        self.reduced = False      # this step's gradient already sits, averaged, in grad_shard


class ShardedAdamW:
    """AdamW whose state is sharded over the data-parallel group."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 group=None, bucket_bytes: int = 64 << 20, overlap: bool = True, direct_grads: bool = True):
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.param_groups = [{"lr": lr}]  # scheduler-compatible view
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("no trainable parameters")
        # autograd produces gradients in reverse registration order; bucket in that
        # order so the first bucket to fill is the first to be reduced
        self.buckets: List[_Bucket] = []
        cur: List[torch.nn.Parameter] = []
        cur_bytes = 0
        for p in reversed(plist):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype):
                self.buckets.append(_Bucket(cur, self.world, self.rank))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(_Bucket(cur, self.world, self.rank))
        # RCCL collectives run on a side stream (default priority).  A gloo group is given host
        # tensors only (shards on a GPU are staged through synchronous host copies) and never overlaps.
        self.backend = dist.get_backend(group) if self.distributed else None
        self.on_rccl = self.distributed and plist[0].is_cuda and self.backend == "nccl"
        self.overlap = overlap and self.on_rccl
        self.comm_stream = torch.cuda.Stream() if self.on_rccl else None
        self._owner: Dict[int, _Bucket] = {}
        # Direct gradients: a bf16 matrix on the GPU gets its slot of the flat buffer published as
        # ``_grad_sink``.  The in-tree wgrad GEMM (``ops/differentiable.py``) writes a step's first
        # gradient straight into the slot and autograd adopts that view as ``.grad``: no zero fill of
        # the buffer and no read-modify-write accumulation into it (for a 7B decoder those are 13.6 GB
        # written plus 41 GB moved per step).  Whatever reaches ``.grad`` some other way is settled into
        # the slot before the buffer is reduced, so correctness never depends on the shortcut.
        self._sink: Dict[int, torch.Tensor] = {}
        self._hooks = []
        for b in self.buckets:
            for p, view in zip(b.params, b.grad_views):
                self._owner[id(p)] = b
                if direct_grads and p.is_cuda and p.dtype == torch.bfloat16 and p.dim() == 2:
                    p._grad_sink = view
                    p._grad_sink_used = True   # until the first zero_grad hands the slot out
                    self._sink[id(p)] = view
                if self.overlap:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._arm()

    def detach(self) -> None:
        """Let go of the parameters: gradient hooks and direct-gradient slots are removed, so another
        optimizer can take the same parameters over (they keep their current values and storage)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                if id(p) in self._sink:
                    del p._grad_sink
                    if hasattr(p, "_grad_sink_used"):
                        del p._grad_sink_used
                p.grad = None
        self._sink = {}

    # ------------------------------------------------------------- hooks / comm
    def _arm(self) -> None:
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            b.reduced = False

    def _settle(self, p: torch.nn.Parameter) -> None:
        """Make the flat buffer hold ``p``'s gradient and ``p.grad`` be its slot again."""
        view = self._sink.get(id(p))
        if view is None:
            return
        g = p.grad
        if g is None:
            view.zero_()                       # no gradient this step
        elif g.data_ptr() != view.data_ptr():
            view.copy_(g)                      # arrived as a tensor of its own
        p.grad = view

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        self._settle(p)
        b = self._owner[id(p)]
        b.pending -= 1
        if b.pending == 0:
            self._launch_reduce(b)

    #: gradient collectives launched so far (tests count them)
    reduces_launched = 0

    def _launch_reduce(self, b: _Bucket) -> None:
        if not self.distributed or self.world == 1:
            b.reduced = True
            return      # grad_shard aliases the flat gradient
        self.reduces_launched += 1
        if self.on_rccl:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                b.work = dist.reduce_scatter_tensor(b.grad_shard, b.flat_grad, op=dist.ReduceOp.AVG,
                                                    group=self.group, async_op=True)
        elif b.flat_grad.is_cuda:                      # gloo: host tensors only
            src = b.flat_grad.cpu()
            dst = torch.empty(b.shard_len, dtype=src.dtype)
            dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.group)
            b.grad_shard.copy_(dst.div_(self.world))
        else:
            dist.reduce_scatter_tensor(b.grad_shard, b.flat_grad, op=dist.ReduceOp.SUM,
                                       group=self.group)
            b.grad_shard.div_(self.world)
        if b.work is None:
            b.reduced = True              # synchronous paths are complete here; async ones once waited for

    def _finish_reduces(self) -> None:
        """Every bucket's averaged gradient shard is in place afterwards.  Idempotent within a step:
        a bucket that has been reduced (by a hook, or by an earlier call -- ``global_grad_norm()``
        before ``step()``) is not reduced again until ``zero_grad()`` re-arms it."""
        if self._sink:
            for b in self.buckets:
                if b.work is None and not b.reduced:
                    for p in b.params:
                        self._settle(p)
        for b in self.buckets:
            if b.reduced:
                continue
            if b.work is None:            # not launched from a hook (no overlap, or a parameter without gradient)
                self._launch_reduce(b)
            if b.work is not None:        # (a one-rank group has nothing in flight)
                b.work.wait()
                b.work = None
                b.reduced = True
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    # -------------------------------------------------------------------- step
    def zero_grad(self, set_to_none: bool = False) -> None:
        for b in self.buckets:
            direct = [id(p) in self._sink for p in b.params]
            if not any(direct):
                b.flat_grad.zero_()
                continue
            for p, view, d in zip(b.params, b.grad_views, direct):
                if d:
                    p.grad = None              # the next gradient is written, not added
                    p._grad_sink_used = False
                else:
                    view.zero_()
        self._arm()

    def _global_sq_norm(self) -> torch.Tensor:
        """Squared global gradient norm as a 0-d fp32 tensor on the shards' device (summed over
        ranks).  Stays on the device with RCCL; a gloo group gets host tensors only."""
        dev = self.buckets[0].grad_shard.device
        if dev.type == "cuda":
            # one streaming pass per bucket (csrc/kernels/optim.hip), accumulated on the device
            from ..ops import hip
            acc = torch.zeros(1, dtype=torch.float32, device=dev)
            for b in self.buckets:
                g = b.grad_shard
                if g.dtype in (torch.bfloat16, torch.float32) and g.data_ptr() % 16 == 0:
                    hip.raw.sumsq_accumulate(g, acc)
                else:
                    acc += torch.linalg.vector_norm(g, dtype=torch.float32).square()
            sq = acc[0]
        else:
            sq = torch.zeros((), dtype=torch.float32, device=dev)
            for b in self.buckets:
                sq += torch.linalg.vector_norm(b.grad_shard, dtype=torch.float32).square()
        if self.distributed and self.world > 1:
            if dist.get_backend(self.group) == "gloo" and sq.is_cuda:
                host = sq.cpu()
                dist.all_reduce(host, group=self.group)
                sq = host.to(sq.device)
            else:
                dist.all_reduce(sq, group=self.group)
        return sq

    def global_grad_norm(self) -> float:
        """Norm of the (averaged) gradient as a host number.  Settles and reduces the buckets
        first, which ``step()`` would do anyway and which is idempotent."""
        self._finish_reduces()
        return float(self._global_sq_norm().sqrt())

    def _gather(self, b: _Bucket, new_shard: torch.Tensor) -> None:
        """All-gather the refreshed shards of one bucket.  On a GPU it goes to the communication
        stream, so the next bucket's AdamW runs while this bucket's shards travel."""
        if self.on_rccl:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                b.work = dist.all_gather_into_tensor(b.flat_param, new_shard, group=self.group, async_op=True)
        elif new_shard.is_cuda:                        # gloo: host tensors only
            full = torch.empty(b.padded, dtype=new_shard.dtype)
            dist.all_gather_into_tensor(full, new_shard.cpu(), group=self.group)
            b.flat_param.copy_(full)
        else:
            dist.all_gather_into_tensor(b.flat_param, new_shard, group=self.group)

    @torch.no_grad()
    def step(self, max_grad_norm: Optional[float] = None) -> torch.Tensor:
        """Reduce-scatter (if not already in flight), clip, AdamW on the shard, all-gather.
        Returns the global gradient norm before clipping as a 0-d tensor: the clip factor is
        computed and consumed on the device (the fused AdamW kernel reads it), so a step on
        RCCL makes no host round trip unless the caller converts the norm (``float(norm)``)."""
        self._finish_reduces()
        norm = self._global_sq_norm().sqrt()
        clip_t = None
        if max_grad_norm is not None:
            clip_t = (max_grad_norm / (norm + 1e-6)).clamp(max=1.0).reshape(1)
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        gather = self.distributed and self.world > 1
        for b in self.buckets:
            b.work = None
            if b.master.is_cuda and b.flat_param.dtype in (torch.bfloat16, torch.float32):
                # one pass over the shard (csrc/kernels/optim.hip) instead of ten launches
                from ..ops import hip
                new_shard = b.new_shard if gather else b.flat_param[b.shard_slice]
                hip.raw.fused_adamw(b.grad_shard, b.master, b.exp_avg, b.exp_avg_sq, new_shard, lr, b1, b2,
                                    self.eps, self.weight_decay, self.step_count, 1.0, clip_t)
                if gather:
                    self._gather(b, new_shard)
                continue
            g = b.grad_shard.float()
            if clip_t is not None:
                g = g * clip_t.to(g.device)
            b.master.mul_(1 - lr * self.weight_decay)
            b.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
            b.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (b.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(self.eps)
            b.master.addcdiv_(b.exp_avg, denom, value=-lr / bc1)
            new_shard = b.master.to(b.flat_param.dtype)
            if gather:
                self._gather(b, new_shard)
            else:
                b.flat_param[b.shard_slice].copy_(new_shard)
        for b in self.buckets:                      # parameters are whole again before the next forward
            if b.work is not None:
                b.work.wait()
                b.work = None
        if self.comm_stream is not None and gather:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        return norm

    # ------------------------------------------------------------- checkpointing
    def state_dict(self) -> dict:
        """This rank's shard of the optimizer state."""
        return {"step": self.step_count, "rank": self.rank, "world": self.world,
                "lr": self.param_groups[0]["lr"],
                "buckets": [{"master": b.master.cpu(), "exp_avg": b.exp_avg.cpu(),
                             "exp_avg_sq": b.exp_avg_sq.cpu()} for b in self.buckets]}

    def load_state_dict(self, sd: dict) -> None:
        if sd["world"] != self.world or len(sd["buckets"]) != len(self.buckets):
            raise ValueError("optimizer shard layout differs from the checkpoint's")
        self.step_count = sd["step"]
        self.param_groups[0]["lr"] = sd.get("lr", self.lr)
        for b, s in zip(self.buckets, sd["buckets"]):
            b.master.copy_(s["master"])
            b.exp_avg.copy_(s["exp_avg"])
            b.exp_avg_sq.copy_(s["exp_avg_sq"])


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s weights."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    on_host = dist.get_backend(group) == "gloo"
    for t in list(module.parameters()) + list(module.buffers()):
        if on_host and t.is_cuda:               # gloo only ever sees host tensors
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=group)
