"""Tensor parallelism for the LLaMA decoder (Megatron-style 1-D sharding).

Sharding over W ranks (one process per GPU, RCCL over xGMI):

    q/k/v     column-parallel by head     rank r owns heads [r*Hq/W, (r+1)*Hq/W)
    o_proj    row-parallel                input columns of those heads
    gate/up   column-parallel             I/W intermediate columns (zero-padded to x64)
    down      row-parallel
    lm_head   vocab-parallel              ceil(V/W) rows; greedy needs only (max, idx)
    norms, embeddings                     replicated

Each layer costs two all-reduces of ``[B, S, H]``.  The residual add is folded
into them: rank 0 accumulates its partial product onto the residual stream,
the other ranks overwrite theirs, and the sum of all ranks' buffers is the new
residual — no extra pass over the activations.

Two executions of the same sharding:
  * ``TensorParallelDecoder`` — nn.Module path through ``eventgpt_amd.ops``; runs
    anywhere (CPU + gloo in the tests) and is the correctness oracle;
  * ``TensorParallelEngine`` — the serving engine (``runtime.engine``) on shards:
    fused GEMV kernels for batch <= 3, split-K MFMA GEMM with fused consumers up to
    32 rows (the bs=32 batched-decode configuration), static KV cache of the local
    heads, all-reduce fused with residual (+ norm), greedy token selection by
    gathering one (max, index) pair per row and rank.

The reference has no model parallelism (its only process group is a
world-size-1 NCCL init demanded by a third-party model,
feasible_imu/benchmark_onellm_5stages.py:393-406).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn

from .. import ops
from ..runtime.engine import GenerationEngine, _Layer, persistent_state


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def shard_layer(lyr, rank: int, world: int, hq: int, hkv: int, d: int, inter: int,
                pad_to: int = 64):
    """Weight shards of one decoder layer for ``rank``.

    Returns ``(w_qkv [(hq_l+2hkv_l)*d, H], w_o [H, hq_l*d], w_gate_up [2*I_l, H],
    w_down [H, I_l])`` with ``I_l`` zero-padded to a multiple of ``pad_to``.
    """
    if hq % world or hkv % world:
        raise ValueError(f"heads ({hq}/{hkv}) must divide by the TP degree {world}")
    hq_l, hkv_l = hq // world, hkv // world
    at, mlp = lyr.self_attn, lyr.mlp
    q = at.q_proj.weight.data[rank * hq_l * d:(rank + 1) * hq_l * d]
    k = at.k_proj.weight.data[rank * hkv_l * d:(rank + 1) * hkv_l * d]
    v = at.v_proj.weight.data[rank * hkv_l * d:(rank + 1) * hkv_l * d]
    w_qkv = torch.cat([q, k, v]).contiguous()
    w_o = at.o_proj.weight.data[:, rank * hq_l * d:(rank + 1) * hq_l * d].contiguous()
    per = (inter + world - 1) // world
    lo, hi = rank * per, min((rank + 1) * per, inter)
    i_l = _round_up(per, pad_to)
    h = mlp.gate_proj.weight.shape[1]
    w_gu = mlp.gate_proj.weight.data.new_zeros(2 * i_l, h)
    w_gu[:hi - lo] = mlp.gate_proj.weight.data[lo:hi]
    w_gu[i_l:i_l + hi - lo] = mlp.up_proj.weight.data[lo:hi]
    w_down = mlp.down_proj.weight.data.new_zeros(h, i_l)
    w_down[:, :hi - lo] = mlp.down_proj.weight.data[:, lo:hi]
    return w_qkv, w_o, w_gu, w_down


def shard_vocab(weight: torch.Tensor, rank: int, world: int) -> Tuple[torch.Tensor, int]:
    """``(rows [ceil(V/W), H] zero-padded, first_row)`` of a vocab-parallel matrix."""
    v = weight.shape[0]
    per = (v + world - 1) // world
    lo, hi = rank * per, min((rank + 1) * per, v)
    out = weight.new_zeros(per, weight.shape[1])
    if hi > lo:
        out[:hi - lo] = weight[lo:hi]
    return out, lo


def _all_reduce(t: torch.Tensor, group) -> torch.Tensor:
    if group is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if dist.get_backend(group) == "gloo" and t.is_cuda:   # gloo only ever sees host tensors
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
        return t
    dist.all_reduce(t, group=group)
    return t


def gather_vocab_shards(local_logits: torch.Tensor, v_total: int, group) -> torch.Tensor:
    """Full ``[B, v_total]`` logits from equally sized (zero/-inf padded) per-rank shards."""
    world = dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1
    if world == 1:
        return local_logits[:, :v_total]
    t = local_logits.contiguous()
    if dist.get_backend(group) == "gloo" and t.is_cuda:   # gloo only ever sees host tensors
        host = t.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host, group=group)
        parts = [b.to(t.device) for b in bufs]
    else:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
    vl = t.shape[1]
    return torch.cat([p[:, :min(vl, max(v_total - r * vl, 0))] for r, p in enumerate(parts)], dim=1)


def distributed_argmax(local_val: torch.Tensor, local_idx: torch.Tensor, group) -> torch.Tensor:
    """Global argmax from per-rank ``(max value, global index)`` pairs ``[B]``; ties go to
    the lowest index.  Only 8 bytes per row per rank cross the fabric."""
    world = dist.get_world_size(group) if (group is not None and dist.is_initialized()) else 1
    if world == 1:
        return local_idx
    pack = torch.stack([local_val.float(), local_idx.float()], dim=-1).contiguous()
    if dist.get_backend(group) == "gloo" and pack.is_cuda:
        host = pack.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host, group=group)
        allp = torch.stack(bufs).to(pack.device)
    else:
        allp = torch.empty(world, *pack.shape, dtype=pack.dtype, device=pack.device)
        dist.all_gather_into_tensor(allp.view(-1, pack.shape[-1]), pack.view(-1, pack.shape[-1]),
                                    group=group)
    vals, idxs = allp[..., 0], allp[..., 1]                # [W, B]
    best = vals.max(dim=0, keepdim=True).values
    cand = torch.where(vals == best, idxs, torch.full_like(idxs, float("inf")))
    return cand.min(dim=0).values.long()


# ------------------------------------------------------------------ module path
class TensorParallelDecoder(nn.Module):
    """The sharded decoder evaluated through ``eventgpt_amd.ops`` (any device)."""

    def __init__(self, model, group=None):
        super().__init__()
        cfg = model.config
        self.cfg, self.group = cfg, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.hq_l = cfg.num_attention_heads // self.world
        self.hkv_l = cfg.num_key_value_heads // self.world
        self.d = cfg.head_dim
        self.shards = [shard_layer(l, self.rank, self.world, cfg.num_attention_heads,
                                   cfg.num_key_value_heads, self.d, cfg.intermediate_size)
                       for l in model.model.layers]
        self.norms = [(l.input_layernorm.weight.data, l.post_attention_layernorm.weight.data)
                      for l in model.model.layers]
        self.final_norm = model.model.norm.weight.data
        self.lm_head, self.v0 = shard_vocab(model.lm_head.weight.data, self.rank, self.world)
        self.v_total = model.lm_head.weight.shape[0]
        self.rope = model.model.rope(model.device)
        self.eps = cfg.rms_norm_eps

    @torch.no_grad()
    def forward(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        """``[B, S, H]`` -> final-norm hidden states ``[B, S, H]`` (identical on all ranks)."""
        x = inputs_embeds
        b, s, _ = x.shape
        pos = torch.arange(s, device=x.device)[None].expand(b, s)
        cos, sin = self.rope
        nq, nk = self.hq_l * self.d, self.hkv_l * self.d
        for (w_qkv, w_o, w_gu, w_down), (n1, n2) in zip(self.shards, self.norms):
            qkv = ops.linear(ops.rmsnorm(x, n1, self.eps), w_qkv)
            q = ops.apply_rope(qkv[..., :nq].reshape(b, s, self.hq_l, self.d), pos, cos, sin)
            k = ops.apply_rope(qkv[..., nq:nq + nk].reshape(b, s, self.hkv_l, self.d), pos, cos, sin)
            v = qkv[..., nq + nk:].reshape(b, s, self.hkv_l, self.d)
            if self.hkv_l != self.hq_l:
                rep = self.hq_l // self.hkv_l
                k, v = k.repeat_interleave(rep, 2), v.repeat_interleave(rep, 2)
            o = ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), causal=True)
            part = ops.linear(o.reshape(b, s, nq), w_o, residual=x if self.rank == 0 else None)
            x = _all_reduce(part, self.group)
            h = ops.swiglu(ops.linear(ops.rmsnorm(x, n2, self.eps), w_gu))
            part = ops.linear(h, w_down, residual=x if self.rank == 0 else None)
            x = _all_reduce(part, self.group)
        return ops.rmsnorm(x, self.final_norm, self.eps)

    @torch.no_grad()
    def greedy_next(self, hidden_last: torch.Tensor) -> torch.Tensor:
        """Greedy token per row of ``hidden_last [B, H]`` via the vocab-parallel head."""
        logits = ops.linear(hidden_last, self.lm_head).float()
        n_valid = min(self.lm_head.shape[0], max(self.v_total - self.v0, 0))
        logits[:, n_valid:] = float("-inf")
        val, idx = logits.max(dim=-1)
        return distributed_argmax(val, idx + self.v0, self.group)


# ------------------------------------------------------------------- engine path
def _host_collective(fn, t: torch.Tensor, *a, **kw):
    """Run a collective on a host copy of ``t`` (gloo only ever sees CPU tensors)."""
    host = t.cpu()
    fn(host, *a, **kw)
    t.copy_(host)
    return t


class TensorParallelEngine(GenerationEngine):
    """``GenerationEngine`` over weight shards, for any batch size.

    ``group=None`` (or a world of one) gives the plain batched engine -- the same code path
    minus the collectives -- which is how it is exercised on one GPU.

    The sharded engine keeps every fusion of the single-GPU one.  Row-parallel projections
    (``o_proj``, ``down_proj``) stop at fp32 partial sums; the collective sits between those
    and the residual add:

      * up to 3 rows: the weight-streaming GEMV writes raw fp32 sums, the all-reduce adds
        the residual into the stream, and the next GEMV's prologue applies the RMSNorm;
      * 4..32 rows: the split-K GEMM leaves its slabs, and the all-reduce launch sums slabs
        and ranks and applies residual + the next RMSNorm (``reduce_residual_norm`` with a
        collective inside);
      * more rows (prefill): MFMA GEMMs with a bf16 all-reduce of the ``[S, H]`` product.

    ``allreduce``: ``"rccl"`` (library collective, the default), ``"oneshot"`` (peer-mapped
    push all-reduce, ``parallel/comm.py``) or ``"auto"`` (one-shot if its start-up self-test
    against RCCL passes).  Greedy selection gathers 8 bytes per row and rank.
    """

    @persistent_state
    def __init__(self, model, group=None, max_len: int = 2048, max_batch: int = 1,
                 allreduce: Optional[str] = None, **kw):
        self._group_arg = group
        self._gloo = bool(group is not None and dist.is_initialized() and dist.get_world_size(group) > 1
                          and dist.get_backend(group) == "gloo")
        if self._gloo:
            kw["use_graph"] = False     # gloo collectives go through host copies: nothing to capture
        if kw.get("weight_format", "bf16") != "bf16" and group is not None and dist.is_initialized() \
                and dist.get_world_size(group) > 1:
            raise NotImplementedError("4-bit weights are not sharded: a tensor-parallel engine of several ranks "
                                      "runs bf16 weights (one rank: up to 3 sequences stream the codes, more "
                                      "multiply by the dequantised matrices)")
        super().__init__(model, max_len=max_len, max_batch=max_batch, **kw)
        dev = self.device
        self.oneshot = None
        self.twoshot = None
        self.allreduce_kind = "none"
        if self.tp_world > 1:
            from . import comm
            # samplers read the gathered, full-vocabulary rows
            self.logits = torch.zeros(self.B, self.v_total, dtype=torch.float32, device=dev)
            self._part = torch.zeros(1, max(self.B, self.gemv_rows), self.H, dtype=torch.float32, device=dev)
            self._pair = torch.zeros(self.B, 2, dtype=torch.int32, device=dev)
            self._pairs = torch.zeros(self.tp_world, self.B, 2, dtype=torch.int32, device=dev)
            mode = (allreduce or comm.allreduce_mode()).lower()
            self.allreduce_kind = "gloo" if self._gloo else "rccl"
            # "oneshot" on a gloo group is the two-processes-on-one-GPU test set-up: gloo only
            # bootstraps (handle exchange, barrier); "auto" needs the library collective to
            # check against, so it is an RCCL-group mode
            if mode == "oneshot" or (mode == "auto" and not self._gloo):
                ar = comm.OneShotAllReduce(self.tp_group, dev)
                if mode == "oneshot" or ar.self_test():
                    self.oneshot, self.allreduce_kind = ar, "oneshot"
                    # the prefill's [rows, hidden] products are bandwidth-bound: two-shot pushes,
                    # sized for a full cache of rows (larger packs take the library collective)
                    rows = min(max(self.l_max, self.PACK_ROWS), 16384)
                    self.twoshot = comm.TwoShotAllReduce(self.tp_group, dev, rows * self.H)
                    # A rank's block waits for the same block of its peers.  Ranks with a GPU each
                    # progress independently; ranks that SHARE one (the gloo-bootstrapped test and
                    # rehearsal set-up) need all their blocks resident together, or the waiting
                    # blocks of one rank keep the others' off the CUs until the waits give up.  There
                    # the two-shot is kept to messages of at most 1536 blocks over all ranks.
                    self._twoshot_cap = 1536 * 2048 if self._gloo else rows * self.H
                    torch.cuda.synchronize(dev)
                    dist.barrier(group=self.tp_group)      # every rank's buffers are mapped before the first push
                else:
                    ar.close()

    # ------------------------------------------------------- collective health / teardown
    def check_collectives(self) -> None:
        """Did a bounded device wait of the in-tree all-reduces (one-shot: decode / verify rows,
        two-shot: prefill) give up?  After a give-up a rank holds other sums than its peers, so
        this is called wherever the host already waits for the device -- the end of a prefill,
        the EOS polls of ``generate``, the batch scheduler's poll -- and the verdict is
        MIN-reduced over the group: every rank raises together instead of drifting apart.

        Skew bound: a wait gives up after ``spin_limit`` polls (2^18, a few hundred ms).  Ranks
        enter a collective further apart than that only when one of them stalls on the host
        (first-launch module load, a garbage collection, an admission only it performs), which is
        why graph capture ends in a barrier and why eager steps are checked here."""
        if self.oneshot is None and self.twoshot is None:
            return
        msg = None
        for ar in (self.oneshot, self.twoshot):
            if ar is None:
                continue
            try:
                ar.check()
            except RuntimeError as e:
                msg = str(e)
                ar.clear_errors()
        ok = torch.tensor([0 if msg else 1], dtype=torch.int32, device="cpu" if self._gloo else self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.tp_group)
        if int(ok.item()) == 0:
            raise RuntimeError("tensor-parallel all-reduce: a device-side wait timed out on "
                               + ("this rank: " + msg if msg else "another rank")
                               + "; the ranks no longer hold the same sums (rebuild the engine, or run with "
                                 "allreduce='rccl')")

    def _poll_hook(self) -> None:
        self.check_collectives()

    def close(self) -> None:
        """Tear the peer mappings down in an order that does not depend on who drops the engine
        first: finish this rank's work, wait until every rank has (nobody pushes into a buffer any
        more), unmap the peers' buffers, wait again (nobody still maps mine), release mine."""
        one, two = self.oneshot, self.twoshot
        self.oneshot = self.twoshot = None
        if one is None and two is None:
            return
        self._graph_greedy.clear()
        self._graph_logits = None
        self._verify_graphs.clear()          # captured launches hold the buffers' addresses
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.tp_group)
        for ar in (one, two):
            if ar is not None:
                ar.close()
        dist.barrier(group=self.tp_group)
        for ar in (one, two):
            if ar is not None:
                ar.buf = None
        self.allreduce_kind = "gloo" if self._gloo else "rccl"

    def _capture_or_eager(self, fn):
        """A captured step is first replayed by ranks that finished capturing at different times.
        The library collective waits for the late ones as long as it takes; the in-tree all-reduce
        waits a bounded number of polls, so with it the ranks leave the capture together."""
        g = super()._capture_or_eager(fn)
        if self.oneshot is not None and self.tp_world > 1:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.tp_group)
        return g

    def _setup_parallel(self) -> None:
        g = self._group_arg
        if g is not None and dist.is_initialized() and dist.get_world_size(g) > 1:
            self.tp_group, self.tp_world, self.tp_rank = g, dist.get_world_size(g), dist.get_rank(g)
        self.Hq_full, self.Hkv_full, self.I_full = self.Hq, self.Hkv, self.I
        self.Hq //= self.tp_world
        self.Hkv //= self.tp_world
        self.I = _round_up((self.I_full + self.tp_world - 1) // self.tp_world, 64)

    def _pack_weights(self, model) -> None:
        bf = dict(dtype=torch.bfloat16, device=self.device)
        for lyr in model.model.layers:
            w_qkv, w_o, w_gu, w_down = shard_layer(lyr, self.tp_rank, self.tp_world, self.Hq_full,
                                                   self.Hkv_full, self.D, self.I_full)
            self.layers.append(_Layer(
                w_qkv, w_o, w_gu, w_down, lyr.input_layernorm.weight.data,
                lyr.post_attention_layernorm.weight.data,
                torch.zeros(self.B, self.Hkv, self.l_max, self.D, **bf),
                torch.zeros(self.B, self.Hkv, self.l_max, self.D, **bf)))
            if self.tp_world > 1:       # the full matrices are no longer needed on this rank
                for p in lyr.parameters():
                    if p.ndim == 2:
                        p.data = p.data.new_empty(0)
        self.final_norm = model.model.norm.weight.data
        shard, self.v0 = shard_vocab(model.lm_head.weight.data, self.tp_rank, self.tp_world)
        self.v_total = model.lm_head.weight.shape[0]
        self.v_per = shard.shape[0]
        # only the rows that exist: the last rank's zero padding never enters an argmax
        self.lm_head = shard[:max(1, min(self.v_per, self.v_total - self.v0))]
        self.embed = model.model.embed_tokens.weight.data

    # -------------------------------------------------------------- collectives
    def _reduce(self, t: torch.Tensor) -> torch.Tensor:
        if (self.twoshot is not None and t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0
                and t.numel() <= self._twoshot_cap and self.twoshot.reduce(t)):
            return t
        return _all_reduce(t, self.tp_group)

    def _allreduce_f32(self, t: torch.Tensor) -> torch.Tensor:
        """Library all-reduce of an fp32 tensor in place."""
        if self._gloo:
            return _host_collective(dist.all_reduce, t, group=self.tp_group)
        dist.all_reduce(t, group=self.tp_group)
        return t

    #: a rank's slice of a row-parallel contraction up to this long is multiplied as ONE slab (the K
    #: split across the ranks already is the split), so the library collective runs on the GEMM's
    #: output in place: projection, all-reduce, residual + norm -- two launches around the collective
    ONE_SLAB_K = 2048

    def _row_partial(self, x, w):
        if self.tp_world > 1 and self.oneshot is None and w.shape[1] <= self.ONE_SLAB_K:
            return self.raw.linear_partial(x, w, 1)
        return self.raw.linear_partial(x, w)

    def _reduce_residual_norm(self, partial, x, norm_w):
        """``partial`` is a fresh ``_row_partial`` result (this engine's to overwrite)."""
        if self.tp_world == 1:
            return super()._reduce_residual_norm(partial, x, norm_w)
        raw = self.raw
        M, N = partial.shape[1], partial.shape[2]
        if self.oneshot is not None:
            x_out = torch.empty(M, N, dtype=torch.bfloat16, device=self.device)
            fuse_norm = norm_w is not None and M > 3 and N <= 4096
            xn = torch.empty_like(x_out) if fuse_norm else None
            if self.oneshot.reduce(partial, res=x.view(M, N), x_out=x_out, norm_w=norm_w if fuse_norm else None,
                                   xn_out=xn, eps=self.eps):
                if norm_w is not None and not fuse_norm:
                    xn = raw.rmsnorm(x_out, norm_w, self.eps)
                return x_out, (xn if norm_w is not None else x_out.new_empty(0))
        # library collective: on the single slab in place, or on the slabs' sum (longer local contractions)
        total = raw.sum_slabs(partial) if partial.shape[0] > 1 else partial
        self._allreduce_f32(total)
        return raw.reduce_residual_norm(total, x.view(M, N), norm_w, self.eps, None)

    def _gemv_residual(self, l, name: str, xin, y) -> None:
        """``y += W[name] @ xin`` summed over the ranks (rows <= 3: weight-streaming GEMV)."""
        if self.tp_world == 1:
            return super()._gemv_residual(l, name, xin, y)
        rows = xin.shape[0]
        part = self._part[:, :rows]
        self.raw.decode_gemv_partial(getattr(l, name), xin, part)
        if self.oneshot is not None and self.oneshot.reduce(part, res=y, x_out=y, eps=self.eps):
            return
        self._allreduce_f32(part)
        self.raw.reduce_residual_norm(part, y, None, self.eps, y)

    # ------------------------------------------------------------------ layers
    def _forward_rows(self, embeds, batch, pos0):
        raw = self.raw
        S = embeds.shape[0]
        if self._gemv_rows_ok(S):
            return self._forward_rows_gemv(embeds, batch, pos0)
        if self._few_rows_ok(S):
            return self._forward_rows_fused(embeds, batch, pos0)
        if self.tp_world == 1:
            return super()._forward_rows(embeds, batch, pos0)      # one rank: the single-GPU prompt path
        x = embeds.to(torch.bfloat16).contiguous().view(1, S, self.H)
        kv_len = pos0 + S
        q = torch.empty(1, S, self.Hq, self.D, dtype=torch.bfloat16, device=self.device)
        o = torch.empty_like(q)
        lead = self.tp_rank == 0
        rope_gemm = self.D == 128 and S <= self.prompt_fused_rows and self.H % 64 == 0
        for l in self.layers:
            kc, vc = l.k_cache[batch:batch + 1], l.v_cache[batch:batch + 1]
            xn = raw.rmsnorm(x, l.norm1, self.eps).view(S, self.H)
            # this rank's heads: RoPE + cache append in the projection's epilogue where the tile allows it
            if not (rope_gemm and raw.linear_qkv_rope(xn, self._w(l, "w_qkv"), q, kc, vc, self.cos, self.sin, pos0,
                                                      self.Hq, self.Hkv)):
                rope_gemm = False
                qkv = raw.linear(xn, self._w(l, "w_qkv"), None, None, 0, False).view(1, S, -1)
                raw.prefill_rope_kv(qkv, q, kc, vc, self.cos, self.sin, pos0, self.Hq, self.Hkv, self.D)
            raw.attention_cache(q, kc, vc, o, kv_len, self.scale)
            x = self._reduce(raw.linear(o.view(1, S, self.Hq * self.D), self._w(l, "w_o"), None,
                                        x if lead else None, 0, False))
            h = self._mlp_up(raw.rmsnorm(x, l.norm2, self.eps), self._w(l, "w_gate_up"))
            x = self._reduce(raw.linear(h, self._w(l, "w_down"), None, x if lead else None, 0, False))
        return x

    #: most rows one packed prefill pass carries (bounds the [rows, 3H] / [rows, 2I] temporaries)
    PACK_ROWS = 8192

    def _forward_rows_packed(self, embeds_list: Sequence[torch.Tensor], slots: Sequence[int]) -> List[torch.Tensor]:
        """Prefill several fresh sequences in ONE pass: the projections run over all prompts' rows
        together (four 636-token prompts are a 2544-row GEMM, which runs the 8-wave kernel, instead
        of four 636-row ones), RoPE / cache append and attention stay per prompt (each slot has its
        own cache rows).  Returns the final hidden state of each prompt's last token."""
        raw = self.raw
        lens = [int(e.shape[0]) for e in embeds_list]
        T = sum(lens)
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        x = torch.cat([e.to(torch.bfloat16) for e in embeds_list]).contiguous().view(1, T, self.H)
        lens_d = torch.tensor(lens, dtype=torch.int32, device=self.device)
        pos0 = torch.zeros(1, dtype=torch.int32, device=self.device)
        q = torch.empty(T, self.Hq, self.D, dtype=torch.bfloat16, device=self.device)
        o = torch.empty_like(q)
        lead = self.tp_rank == 0
        for l in self.layers:
            qkv = raw.linear(raw.rmsnorm(x, l.norm1, self.eps), self._w(l, "w_qkv"), None, None, 0, False).view(T, -1)
            for i, b in enumerate(slots):
                a, e = offs[i], offs[i + 1]
                kc, vc = l.k_cache[b:b + 1], l.v_cache[b:b + 1]
                qi, oi = q[a:e].view(1, e - a, self.Hq, self.D), o[a:e].view(1, e - a, self.Hq, self.D)
                raw.prefill_rope_kv(qkv[a:e].view(1, e - a, -1), qi, kc, vc, self.cos, self.sin, pos0,
                                    self.Hq, self.Hkv, self.D)
                raw.attention_cache(qi, kc, vc, oi, lens_d[i:i + 1], self.scale)
            x = self._reduce(raw.linear(o.view(1, T, self.Hq * self.D), self._w(l, "w_o"), None, x if lead else None, 0, False))
            h = self._mlp_up(raw.rmsnorm(x, l.norm2, self.eps), self._w(l, "w_gate_up"))
            x = self._reduce(raw.linear(h, self._w(l, "w_down"), None, x if lead else None, 0, False))
        return [x[0, offs[i + 1] - 1] for i in range(len(slots))]

    def _prefill_slots(self, slots: Sequence[int], embeds_list: Sequence[torch.Tensor]) -> None:
        """Fill the caches of ``slots`` from their prompts and stage each last hidden state for the
        commit: prompts are packed into passes of at most ``PACK_ROWS`` rows; a pass of one prompt
        (or prompts short enough for the few-row kernels) takes the single-sequence path."""
        for emb in embeds_list:
            if emb.shape[0] > self.l_max:
                raise ValueError(f"prompt of {emb.shape[0]} tokens exceeds the cache ({self.l_max})")
        i, n = 0, len(slots)
        while i < n:
            j, rows = i, 0
            while j < n and (j == i or rows + embeds_list[j].shape[0] <= self.PACK_ROWS):
                rows += int(embeds_list[j].shape[0])
                j += 1
            group_s, group_e = list(slots[i:j]), list(embeds_list[i:j])
            if j - i >= 2 and min(int(e.shape[0]) for e in group_e) > 32:
                lasts = self._forward_rows_packed(group_e, group_s)
            else:
                lasts = []
                for b, emb in zip(group_s, group_e):
                    pos0 = torch.zeros(1, dtype=torch.int32, device=self.device)
                    lasts.append(self._forward_rows(emb, b, pos0)[0, -1])
            for b, emb, last in zip(group_s, group_e, lasts):
                self.x[b].copy_(last)
                self.pos[b] = emb.shape[0] - 1      # the commit advances it to the prompt length
                self._pos_hi = max(self._pos_hi, int(emb.shape[0]))   # host bound behind cache_room()
            i = j

    def _layers_step(self) -> None:
        raw = self.raw
        if self.B <= self.gemv_rows:     # measured: 2 / 3 / 4 streams 2.79 / 3.06 / 3.39 ms per step, 4 as a GEMM 3.61
            return super()._layers_step()
        # batched decode (e.g. 32 streams): M = B rows through the MFMA GEMM
        B = self.B
        x = self.x
        if self._few_rows_ok(B):
            # <= 32 streams: the split-K consumers carry RoPE/cache append, residual + next
            # norm and SwiGLU (no separate launches for them); with several ranks the
            # residual consumer is also the all-reduce
            n = len(self.layers)
            xn = raw.rmsnorm(x, self.layers[0].norm1, self.eps)
            for i, l in enumerate(self.layers):
                raw.reduce_rope_kv(raw.linear_partial(xn, self._w(l, "w_qkv")), self.q, l.k_cache, l.v_cache,
                                   self.cos, self.sin, self.pos, B, 1, self.Hq, self.Hkv, self.D)
                raw.decode_attn(self.q, l.k_cache, l.v_cache, self.attn, self.part_o, self.part_ml,
                                self.pos, self.nsplit, self.scale)
                x, xn = self._reduce_residual_norm(self._row_partial(self.attn, self._w(l, "w_o")), x, l.norm2)
                h = raw.reduce_swiglu(raw.linear_partial(xn, self._w(l, "w_gate_up")))
                nxt = self.layers[i + 1].norm1 if i + 1 < n else None
                x, xn = self._reduce_residual_norm(self._row_partial(h, self._w(l, "w_down")), x, nxt)
            self.x.copy_(x)
            return
        lead = self.tp_rank == 0
        for l in self.layers:
            qkv = raw.linear(raw.rmsnorm(x, l.norm1, self.eps), self._w(l, "w_qkv"), None, None, 0, False)
            raw.prefill_rope_kv(qkv.view(B, 1, -1), self.q.view(B, 1, -1), l.k_cache, l.v_cache,
                                self.cos, self.sin, self.pos, self.Hq, self.Hkv, self.D)
            raw.decode_attn(self.q, l.k_cache, l.v_cache, self.attn, self.part_o, self.part_ml,
                            self.pos, self.nsplit, self.scale)
            x = self._reduce(raw.linear(self.attn, self._w(l, "w_o"), None, x if lead else None, 0, False))
            h = self._mlp_up(raw.rmsnorm(x, l.norm2, self.eps), self._w(l, "w_gate_up"))
            x = self._reduce(raw.linear(h, self._w(l, "w_down"), None, x if lead else None, 0, False))
        self.x.copy_(x)

    # ------------------------------------------------------------------ head
    def _gather_pairs(self) -> None:
        """``self._pair [B,2]`` of every rank -> ``self._pairs [W,B,2]`` (8 bytes per row and rank)."""
        if self._gloo:
            host = self._pair.cpu()
            bufs = [torch.empty_like(host) for _ in range(self.tp_world)]
            dist.all_gather(bufs, host, group=self.tp_group)
            self._pairs.copy_(torch.stack(bufs))
        else:
            dist.all_gather_into_tensor(self._pairs.view(-1), self._pair.view(-1), group=self.tp_group)

    def _logits_and_commit(self, greedy: bool, want_logits: bool = False) -> None:
        raw = self.raw
        full = (not greedy) or want_logits
        if self.tp_world == 1 and self.B <= int(raw.gemv_max_batch()):
            return super()._logits_and_commit(greedy, want_logits)
        if full:
            hn = raw.rmsnorm(self.x, self.final_norm, self.eps)
            local = raw.linear(hn, self.lm_head, None, None, 0, True)           # fp32 [B, V_local]
            if self.tp_world > 1:
                pad = self.v_per - local.shape[1]
                if pad:
                    local = torch.nn.functional.pad(local, (0, pad), value=float("-inf"))
                self.logits.copy_(gather_vocab_shards(local, self.v_total, self.tp_group))
            else:
                self.logits.copy_(local[:, :self.logits.shape[1]])
        if not greedy:
            return
        # per-row argmax partials of this rank's rows, never the logits themselves
        if self.B <= int(raw.gemv_max_batch()):
            raw.decode_logits(self.lm_head, self.x, self.final_norm, self.eps, None, self.part_val, self.part_idx)
            pv, pi, n_part = self.part_val, self.part_idx, self.n_part
        else:
            hn = raw.rmsnorm(self.x, self.final_norm, self.eps)
            pv, pi = raw.linear_argmax_partials(hn, self.lm_head, None)
            n_part = pv.shape[1]
        if self.tp_world == 1:
            raw.commit_token(pv, pi, n_part, None, self.embed, self.x, self.out_tokens, self.step, self.pos,
                             self.done, self.cur_token, self.eos, self.active)
            return
        raw.argmax_pair_local(pv, pi, self.B, n_part, self.v0, self._pair)
        self._gather_pairs()
        raw.argmax_pair_select(self._pairs, self.tp_world, self.B, self.sampled)
        self._commit_sampled()

    # ----------------------------------------------------------------- batching
    @torch.no_grad()
    def prefill_batch(self, embeds_list: Sequence[torch.Tensor]) -> None:
        """Prefill sequence ``b`` with ``embeds_list[b]`` for every slot, then pick and
        commit the first token of all of them in one step."""
        if len(embeds_list) != self.B:
            raise ValueError(f"need {self.B} prompts, got {len(embeds_list)}")
        self.reset()
        self._prefill_slots(list(range(self.B)), list(embeds_list))
        self._logits_and_commit(greedy=True)
        self._after_prefill()

    def _after_prefill(self) -> None:
        """The in-tree collectives of a prefill are checked once its work is done (one device sync per
        prompt, where the library collective would have synchronised the ranks anyway)."""
        if self.oneshot is not None or self.twoshot is not None:
            torch.cuda.synchronize(self.device)
            self.check_collectives()

    def prefill(self, embeds, batch: int = 0, start: Optional[int] = None,
                select: Optional[str] = "greedy", return_hidden: bool = False):
        if self.B == 1:
            out = super().prefill(embeds, batch, start, select, return_hidden)
        else:
            if select is not None:
                raise ValueError("with several sequences use prefill_batch (one commit for all)")
            out = super().prefill(embeds, batch, start, None, return_hidden)
        self._after_prefill()
        return out

    # ------------------------------------------------------ slot-wise scheduling
    def enable_slot_masks(self) -> None:
        """Give the commit kernel a per-slot "may advance" mask (all ones).  Captured
        graphs bake the pointer, so earlier captures are dropped."""
        if self.active is None:
            self.active = torch.ones(self.B, dtype=torch.int32, device=self.device)
            self._graph_greedy.clear()
            self._graph_logits = None

    @torch.no_grad()
    def admit(self, slots: Sequence[int], embeds_list: Sequence[torch.Tensor]) -> None:
        """Start new sequences in ``slots`` while the other slots keep their state:
        prefill each slot's cache, then pick and commit the first token of exactly those
        slots (every other slot is masked out of the commit)."""
        self.enable_slot_masks()
        if len(slots) != len(embeds_list) or not slots:
            raise ValueError("one prompt per admitted slot")
        keep = self.active.clone()
        self._prefill_slots(list(slots), list(embeds_list))
        self.step[list(slots)] = 0
        self.done[list(slots)] = 0
        self.active.zero_()
        self.active[list(slots)] = 1
        self._logits_and_commit(greedy=True)
        keep[list(slots)] = 1
        self.active.copy_(keep)
        self._after_prefill()

    def park(self, slots: Sequence[int]) -> None:
        """Stop advancing ``slots`` (finished or cancelled sequences)."""
        self.enable_slot_masks()
        if slots:
            self.active[list(slots)] = 0
            # a parked slot still rides through the batched kernels: with position 0 its
            # attention reads one key instead of a finished sequence's whole cache
            self.pos[list(slots)] = 0

    def tokens(self, batch: int) -> torch.Tensor:
        n = int(self.step[batch].item())
        return self.out_tokens[batch, :n].to(torch.long)
