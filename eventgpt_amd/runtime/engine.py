"""GenerationEngine — the MI355X serving path of the decoder.

What it is: a static-memory executor for prefill + autoregressive decode of a
LLaMA-family decoder whose weights live in an ``EventChatModel``.

  * weights are re-packed once (``q|k|v`` rows in one matrix, ``gate|up`` rows
    in one matrix); the module's parameters are re-pointed at views of the
    packed storage, so there is one copy in HBM;
  * the KV cache is pre-allocated ``[layers][B, Hkv, Lmax, D]``; sequence
    positions, the step counter, EOS flags and the token ring live in device
    memory and are advanced by the kernels themselves;
  * one decode step is 6 launches per layer (fused norm+QKV+RoPE+cache append,
    split-KV attention, combine, o_proj+residual, fused norm+gate/up+SwiGLU,
    down+residual) plus lm_head/argmax and a token-commit kernel that also
    gathers the next embedding — no host work between steps;
  * the whole step is captured once into a HIP graph and replayed per token;
    the host only polls the EOS flag every ``poll_every`` tokens.

``weight_format="nf4"`` (or ``"fp4"``) keeps the four projection matrices of every
layer as 4-bit codes + one fp32 scale per 64 weights (``eventgpt_amd.quant``): the
decode GEMVs stream a quarter of the bytes and look the levels up in registers;
passes with more rows run the MFMA GEMM on the dequantised matrix — kept in HBM in
place of the original bf16 weights (so the module path computes the same quantised
model), or, with ``release_bf16``, expanded one matrix at a time into a bf16 scratch.  The LM head and the embedding stay bf16, as in the reference's
bitsandbytes configuration (pipeline/feature_extraction/extract_hidden_states.py:117-122).

Prefill runs the same weights through the MFMA GEMM and the fused causal
attention kernel and appends to the same cache.  The reference's equivalent is
HF ``generate`` / its hand-rolled loops with one host sync and several
allocations per token (pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:417-472).
"""
from __future__ import annotations

import contextlib
import dataclasses
import functools
import gc
import math
from typing import List, Optional

import torch

from .. import ops


@dataclasses.dataclass
class _Layer:
    w_qkv: torch.Tensor
    w_o: torch.Tensor
    w_gate_up: torch.Tensor
    w_down: torch.Tensor
    norm1: torch.Tensor
    norm2: torch.Tensor
    k_cache: torch.Tensor
    v_cache: torch.Tensor
    q4: Optional[dict] = None     # name -> W4Weight when the engine runs 4-bit weights


def _pad_rows(w: torch.Tensor, mult: int) -> torch.Tensor:
    pad = (-w.shape[0]) % mult
    if pad == 0:
        return w
    return torch.cat([w, w.new_zeros(pad, w.shape[1])])


@contextlib.contextmanager
def capture_tensor_mode():
    """Tensor mode for a graph capture: plain ``no_grad``, also inside a caller's ``torch.inference_mode()``.
    A capture allocates tensors that outlive it (the graph's static outputs, and the seed / offset
    tensors torch attaches to the default generator while any graph is alive); allocated under
    inference mode they are inference tensors, and the next capture or replay bookkeeping that updates
    them outside inference mode raises and leaves the generator stuck in its capturing state."""
    with torch.inference_mode(False), torch.no_grad():
        yield


@contextlib.contextmanager
def no_gc_during_capture():
    """Keep Python's cyclic garbage collector from running inside a stream capture.  A collection that
    happens to free an older engine's ``CUDAGraph`` (or anything else whose destructor calls into the HIP
    runtime) while this thread is capturing makes that call fail, and a failure inside a destructor ends the
    process.  ``torch.cuda.graph`` collects once before the capture begins; this covers the capture itself."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def persistent_state(method):
    """Decorator for methods that allocate tensors an engine keeps and later updates in place (its
    state, caches, static graph inputs): they run in ``capture_tensor_mode`` so that an engine built or
    first used under ``torch.inference_mode()`` still works outside it."""
    @functools.wraps(method)
    def wrapped(*args, **kwargs):
        with capture_tensor_mode():
            return method(*args, **kwargs)
    return wrapped


class GenerationEngine:
    @persistent_state
    def __init__(self, model, max_len: int = 2560, max_batch: int = 1, nsplit: Optional[int] = None,
                 use_graph: bool = True, max_new_tokens: int = 1024, poll_every: int = 16,
                 weight_format: str = "bf16", release_bf16: bool = False):
        from ..ops import hip  # raises if the extension is missing
        self.raw = hip.raw
        cfg = model.config
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError("GenerationEngine needs the model on a GPU")
        if model.dtype != torch.bfloat16:
            raise RuntimeError("GenerationEngine runs bf16 weights")
        self.model = model
        self.cfg = cfg
        self.device = dev
        self.B = max_batch
        self.H = cfg.hidden_size
        self.Hq = cfg.num_attention_heads
        self.Hkv = cfg.num_key_value_heads
        self.D = cfg.head_dim
        self.I = cfg.intermediate_size
        self.eps = float(cfg.rms_norm_eps)
        self.l_max = max_len
        self._nsplit_arg = nsplit
        self.use_graph = use_graph
        self.max_out = max_new_tokens
        self.poll_every = poll_every
        self.scale = 1.0 / math.sqrt(self.D)
        if self.D not in (64, 128):
            raise ValueError("head_dim must be 64 or 128 for the HIP attention kernels")
        if self.H % 64 or self.I % 8:
            raise ValueError("hidden size must be a multiple of 64, intermediate of 8")

        if weight_format not in ("bf16", "nf4", "fp4"):
            raise ValueError(f"weight_format {weight_format!r}: bf16, nf4 or fp4")
        self.weight_format = weight_format
        self.w4 = weight_format != "bf16"
        self.tp_world, self.tp_rank, self.tp_group = 1, 0, None
        self._setup_parallel()
        if self.w4 and self.tp_world > 1:
            raise NotImplementedError("4-bit weights are implemented for the single-GPU engine only")
        self.layers: List[_Layer] = []
        self._pack_weights(model)
        if self.w4:
            self._quantize_layers(release_bf16)
        self.V = self.lm_head.shape[0]
        cos, sin = ops.rope_tables(max_len, self.D, cfg.rope_theta, dev)
        self.cos, self.sin = cos.contiguous(), sin.contiguous()

        B = self.B
        nsplit = self.nsplit = self.kv_splits(B, self.Hq) if self._nsplit_arg is None else int(self._nsplit_arg)
        bf = dict(dtype=torch.bfloat16, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.x = torch.zeros(B, self.H, **bf)           # residual stream of the decode step
        self.q = torch.zeros(B, self.Hq * self.D, **bf)
        self.attn = torch.zeros(B, self.Hq * self.D, **bf)
        self.h = torch.zeros(B, self.I, **bf)
        self.part_o = torch.zeros(B * self.Hq * nsplit * self.D, **f32)
        self.part_ml = torch.zeros(B * self.Hq * nsplit * 2, **f32)
        self.n_part = int(self.raw.logits_partials(self.V))
        self.part_val = torch.zeros(B * self.n_part, **f32)
        self.part_idx = torch.zeros(B * self.n_part, **i32)
        self.logits = torch.zeros(B, self.V, **f32)
        self.out_tokens = torch.zeros(B, self.max_out, **i32)
        self.step = torch.zeros(B, **i32)
        self.pos = torch.zeros(B, **i32)
        self.done = torch.zeros(B, **i32)
        self.cur_token = torch.zeros(B, **i32)
        self.sampled = torch.zeros(B, **i32)
        self.sd_result = torch.zeros(8, **i32)  # [n_accepted, next_token, n_emitted, step, done, pos, -, -]
        self.sd_limit = None     # device int32 [1]: freeze verify rounds once this many tokens are out
        self.sd_stats = None     # device int32 [4]: drafted, accepted, rounds (device-resident SD loops)
        # per-sequence "may advance" flags read by the commit kernel; None = all.  A batch
        # scheduler allocates them (TensorParallelEngine.enable_slot_masks).
        self.active = None
        self.eos = int(cfg.eos_token_id if cfg.eos_token_id is not None else -1)
        # captured graphs bake the EOS id (a launch argument), so they are keyed by it
        self._graph_greedy = {}
        self._graph_logits = None
        self._verify_graphs = {}

    # ---------------------------------------------------------------- weights
    def _setup_parallel(self) -> None:
        """Hook for tensor-parallel subclasses (sets tp_* and local head counts)."""

    def _pack_weights(self, model) -> None:
        """Pack ``q|k|v`` and ``gate|up`` rows and allocate the KV cache.  The module's
        parameters are re-pointed at views of the packed storage (one copy in HBM)."""
        bf = dict(dtype=torch.bfloat16, device=self.device)
        with torch.no_grad():
            for lyr in model.model.layers:
                at, mlp = lyr.self_attn, lyr.mlp
                w_qkv = torch.cat([at.q_proj.weight.data, at.k_proj.weight.data,
                                   at.v_proj.weight.data]).contiguous()
                nq, nk = self.Hq * self.D, self.Hkv * self.D
                at.q_proj.weight.data = w_qkv[:nq]
                at.k_proj.weight.data = w_qkv[nq:nq + nk]
                at.v_proj.weight.data = w_qkv[nq + nk:]
                w_gu = torch.cat([mlp.gate_proj.weight.data, mlp.up_proj.weight.data]).contiguous()
                mlp.gate_proj.weight.data = w_gu[:self.I]
                mlp.up_proj.weight.data = w_gu[self.I:]
                self.layers.append(_Layer(
                    w_qkv, at.o_proj.weight.data, w_gu, mlp.down_proj.weight.data,
                    lyr.input_layernorm.weight.data, lyr.post_attention_layernorm.weight.data,
                    torch.zeros(self.B, self.Hkv, self.l_max, self.D, **bf),
                    torch.zeros(self.B, self.Hkv, self.l_max, self.D, **bf)))
        self.final_norm = model.model.norm.weight.data
        self.lm_head = model.lm_head.weight.data
        self.embed = model.model.embed_tokens.weight.data

    _W4_NAMES = ("w_qkv", "w_o", "w_gate_up", "w_down")

    def _quantize_layers(self, release_bf16: bool) -> None:
        """Quantise the packed projections of every layer (on the GPU, ~1 s for 7B).

        By default the bf16 weights are then overwritten IN PLACE by their dequantised
        values: multi-row passes (prefill, long verifies) run at bf16-GEMM speed on exactly
        the quantised model, and the module path (whose parameters are views of the same
        storage) computes that model too.  With ``release_bf16`` the bf16 copies are dropped
        instead (the module path stops working; the engine then holds 3.9 GB of weights for a
        7B decoder instead of 13 GB) and multi-row passes expand one matrix at a time into a
        scratch buffer per projection kind."""
        from ..quant import quantize_weight
        if self.H % 64 or self.I % 64 or (self.Hq * self.D) % 64:
            raise ValueError("4-bit weights need every projection's input size to be a multiple of 64")
        self._w4_scratch = {}
        self._w4_release = bool(release_bf16)
        for l in self.layers:
            l.q4 = {n: quantize_weight(getattr(l, n), self.weight_format) for n in self._W4_NAMES}
            for n in self._W4_NAMES:
                if release_bf16:
                    if n not in self._w4_scratch:
                        self._w4_scratch[n] = torch.empty_like(getattr(l, n))
                else:
                    l.q4[n].dequantize(getattr(l, n))
        if release_bf16:
            empty = torch.empty(0, dtype=torch.bfloat16, device=self.device)
            for l, lyr in zip(self.layers, self.model.model.layers):
                for n in self._W4_NAMES:
                    setattr(l, n, None)
                for lin in (lyr.self_attn.q_proj, lyr.self_attn.k_proj, lyr.self_attn.v_proj,
                            lyr.self_attn.o_proj, lyr.mlp.gate_proj, lyr.mlp.up_proj, lyr.mlp.down_proj):
                    lin.weight.data = empty
            torch.cuda.empty_cache()

    def _w(self, l: _Layer, name: str) -> torch.Tensor:
        """The bf16 matrix a GEMM path multiplies by: the weight itself, or the 4-bit
        weight expanded into this projection kind's scratch."""
        if not self.w4 or not self._w4_release:
            return getattr(l, name)
        return l.q4[name].dequantize(self._w4_scratch[name])

    @property
    def weight_bytes(self) -> int:
        """Bytes of decoder-layer weights one decode step streams."""
        if self.w4:
            return sum(q.nbytes for l in self.layers for q in l.q4.values())
        return sum(getattr(l, n).numel() * 2 for l in self.layers for n in self._W4_NAMES)

    # ------------------------------------------------------------------ state
    def reset(self) -> None:
        for t in (self.step, self.pos, self.done, self.cur_token):
            t.zero_()
        self._pos_hi = 0

    # Host-side upper bound of the device position counters.  The kernels clamp a position that
    # runs past the cache (nothing faults), but a clamped verify silently accepts or emits wrong
    # tokens, so multi-row passes are refused on the host before they can overrun: the bound is
    # advanced by the most a call can append and re-read exactly (one small copy) only when it
    # says the cache might be full.
    _pos_hi = 0

    def cache_room(self, rows: int) -> bool:
        """True if ``rows`` more positions fit the cache of every sequence."""
        if self._pos_hi + rows > self.l_max:
            self._pos_hi = int(self.pos.max().item())
        return self._pos_hi + rows <= self.l_max

    @property
    def kv_bytes_per_token(self) -> int:
        return 2 * len(self.layers) * self.Hkv * self.D * 2

    def past_key_values(self, batch: int = 0):
        """HF-style tuple of ``(k, v)`` views ``[1, Hkv, L, D]`` of the live cache."""
        n = int(self.pos[batch].item())
        return tuple((l.k_cache[batch:batch + 1, :, :n], l.v_cache[batch:batch + 1, :, :n])
                     for l in self.layers)

    # ---------------------------------------------------------------- prefill
    @torch.no_grad()
    def prefill(self, embeds: torch.Tensor, batch: int = 0, start: Optional[int] = None,
                select: Optional[str] = "greedy", return_hidden: bool = False):
        """Run ``embeds [S, H]`` through the decoder, appending to sequence ``batch``'s
        cache from position ``start`` (default: its current length).

        ``select="greedy"`` picks the next token from the last row and commits it
        exactly as a decode step would; ``"logits"`` leaves fp32 logits of the last
        row in ``self.logits`` for a sampler (commit with ``_commit_sampled``);
        ``None`` only fills the cache.  Returns the last-layer hidden states
        ``[S, H]`` (before the final norm) if ``return_hidden``.
        """
        S = embeds.shape[0]
        if start is None:
            start = int(self.pos[batch].item())
        if start + S > self.l_max:
            raise ValueError(f"sequence of {start + S} tokens exceeds the cache ({self.l_max})")
        pos0 = torch.tensor([start], dtype=torch.int32, device=self.device)
        x = self._forward_rows(embeds, batch, pos0)
        self._pos_hi = max(self._pos_hi, start + S)
        if select is None:
            self.pos[batch] = start + S
        else:
            # the commit kernel advances pos by one, like a decode step does
            self.pos[batch] = start + S - 1
            self.x[batch].copy_(x[0, -1])
            self._logits_and_commit(greedy=(select == "greedy"))
        return x[0] if return_hidden else None

    # Up to gemv_rows rows of ONE sequence (short speculative verifies) run on the decode
    # step's weight-streaming kernels (7B, 2 115 keys, ms per verify pass of 1 / 2 / 3 drafts: 3.03 / 3.23 /
    # 3.47, the few-row GEMM path 3.16* / 3.49* / 3.67 -- * on the multi-row GEMV that preceded the rows kernel): fused norm+QKV+RoPE+cache append, o_proj+residual,
    # fused norm+gate/up+SwiGLU, down+residual: 4 launches + attention per layer.
    gemv_rows = 4

    @staticmethod
    def kv_splits(streams: int, heads: int) -> int:
        """KV splits of the decode attention (one block per stream, head and split): 16 for one or two
        streams; from three on the power of two that makes about 512 blocks, and one split -- the block writes
        the output row itself, no partial results and no merge launch -- once streams x heads fill the CUs.
        7B decode step at a 636-token context, ms, 16 splits against this choice: 4 streams 3.35 / 3.30,
        8 streams 3.95 / 3.81, 16 streams 4.48 / 4.33, 32 streams 5.66 / 5.49 (two streams: 8 splits 2.766,
        16 splits 2.755; four streams with one split 3.36)."""
        if streams <= 2:
            return 16
        blocks = max(1, streams * heads)
        if blocks >= 256:
            return 1
        want = 512 // blocks
        return min(16, 1 << (want.bit_length() - 1))

    def _gemv_rows_ok(self, rows: int) -> bool:
        cap = int(self.raw.gemv_w4_max_batch() if self.w4 else self.raw.gemv_max_batch())
        return 1 <= rows <= min(self.gemv_rows, cap)

    def _forward_rows_gemv(self, embeds: torch.Tensor, batch: int, pos0: torch.Tensor) -> torch.Tensor:
        raw = self.raw
        S = embeds.shape[0]
        x = embeds.to(torch.bfloat16).reshape(S, self.H).clone()      # residual stream, updated in place
        kv_len = pos0 + S
        q = torch.empty(1, S, self.Hq, self.D, dtype=torch.bfloat16, device=self.device)
        o = torch.empty_like(q)
        h = torch.empty(S, self.I, dtype=torch.bfloat16, device=self.device)
        for l in self.layers:
            kc, vc = l.k_cache[batch:batch + 1], l.v_cache[batch:batch + 1]
            self._gemv_qkv(l, x, q, kc, vc, pos0, True)
            raw.attention_cache(q, kc, vc, o, kv_len, self.scale)
            self._gemv_residual(l, "w_o", o.view(S, self.Hq * self.D), x)
            self._gemv_mlp_up(l, x, h)
            self._gemv_residual(l, "w_down", h, x)
        return x.view(1, S, self.H)

    # the four weight-streaming launches of a layer, in either weight format
    def _w4_rows(self, rows: int) -> bool:
        """4-bit weights are streamed as codes for up to ``gemv_w4_max_batch`` rows; a step of more
        sequences multiplies by the dequantised bf16 matrices (``_w``), as every multi-row pass does:
        the same quantised model, at the bf16 kernels' speed."""
        return self.w4 and rows <= int(self.raw.gemv_w4_max_batch())

    def _gemv_qkv(self, l, x, q, kc, vc, pos, one_seq: bool) -> None:
        if self._w4_rows(x.shape[0]):
            w = l.q4["w_qkv"]
            self.raw.decode_qkv_w4(x, l.norm1, w.packed, w.absmax, w.code, q, kc, vc, self.cos, self.sin,
                                   pos, self.eps, self.Hq, self.Hkv, self.D, one_seq)
        else:
            self.raw.decode_qkv(x, l.norm1, self._w(l, "w_qkv"), q, kc, vc, self.cos, self.sin, pos, self.eps,
                                self.Hq, self.Hkv, self.D, one_seq)

    def _gemv_residual(self, l, name: str, xin, y) -> None:
        if self._w4_rows(xin.shape[0]):
            w = l.q4[name]
            self.raw.decode_gemv_residual_w4(w.packed, w.absmax, w.code, xin, y)
        else:
            self.raw.decode_gemv_residual(self._w(l, name), xin, y)

    def _gemv_mlp_up(self, l, x, h) -> None:
        if self._w4_rows(x.shape[0]):
            w = l.q4["w_gate_up"]
            self.raw.decode_mlp_up_w4(w.packed, w.absmax, w.code, x, l.norm2, self.eps, h)
        else:
            self.raw.decode_mlp_up(self._w(l, "w_gate_up"), x, l.norm2, self.eps, h)

    # Row counts up to this take the few-row path: projections stop at their fp32
    # split-K slabs and the op that follows (RoPE + cache append, residual + next
    # RMSNorm, SwiGLU) is done by the kernel that sums the slabs.
    fused_rows = 32

    def _mlp_up(self, xn: torch.Tensor, w_gate_up: torch.Tensor) -> torch.Tensor:
        """``silu(xn Wgate^T) * (xn Wup^T)``: above 32 rows the GEMM applies the activation in its
        epilogue (the ``[rows, 2I]`` intermediate is never written); below, GEMM + activation kernel."""
        rows = xn.numel() // xn.shape[-1]
        if rows > 32 and (w_gate_up.shape[0] // 2) % 32 == 0 and xn.shape[-1] % 64 == 0:
            return self.raw.linear_swiglu(xn, w_gate_up)
        return self.raw.swiglu(self.raw.linear(xn, w_gate_up, None, None, 0, False))

    def _few_rows_ok(self, rows: int) -> bool:
        return (3 < rows <= self.fused_rows and self.H % 64 == 0
                and self.I % 64 == 0 and (self.Hq * self.D) % 64 == 0)

    def _row_partial(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """fp32 split-K slabs ``[nslab, M, N]`` of a row-parallel projection (``o_proj`` / ``down_proj``)
        for ``_reduce_residual_norm``."""
        return self.raw.linear_partial(x, w)

    def _reduce_residual_norm(self, partial, x, norm_w):
        """``(x + bf16(sum of the split-K slabs), RMSNorm of that * norm_w)`` of a row-parallel
        projection's partials ``[nslab, M, N]``.  A tensor-parallel engine also sums over the
        ranks here: the collective sits between the slab sum and the residual add."""
        return self.raw.reduce_residual_norm(partial, x, norm_w, self.eps, None)

    def _forward_rows_fused(self, embeds: torch.Tensor, batch: int, pos0: torch.Tensor) -> torch.Tensor:
        raw = self.raw
        S = embeds.shape[0]
        x = embeds.to(torch.bfloat16).contiguous().view(S, self.H)
        kv_len = pos0 + S
        q = torch.empty(1, S, self.Hq, self.D, dtype=torch.bfloat16, device=self.device)
        o = torch.empty_like(q)
        xn = raw.rmsnorm(x, self.layers[0].norm1, self.eps)
        n = len(self.layers)
        for i, l in enumerate(self.layers):
            kc, vc = l.k_cache[batch:batch + 1], l.v_cache[batch:batch + 1]
            raw.reduce_rope_kv(raw.linear_partial(xn, self._w(l, "w_qkv")), q, kc, vc, self.cos, self.sin, pos0,
                               1, S, self.Hq, self.Hkv, self.D)
            raw.attention_cache(q, kc, vc, o, kv_len, self.scale)
            x, xn = self._reduce_residual_norm(self._row_partial(o.view(S, self.Hq * self.D), self._w(l, "w_o")),
                                               x, l.norm2)
            h = raw.reduce_swiglu(raw.linear_partial(xn, self._w(l, "w_gate_up")))
            nxt = self.layers[i + 1].norm1 if i + 1 < n else None
            x, xn = self._reduce_residual_norm(self._row_partial(h, self._w(l, "w_down")), x, nxt)
        return x.view(1, S, self.H)

    def _forward_rows(self, embeds: torch.Tensor, batch: int, pos0: torch.Tensor) -> torch.Tensor:
        """All decoder layers over ``embeds [S, H]`` placed at cache positions
        ``pos0[0] .. pos0[0]+S-1`` of sequence ``batch`` (``pos0`` is a device int32
        tensor, so no host sync).  Returns hidden states ``[1, S, H]`` before the
        final norm.  Does not touch the position counters."""
        raw = self.raw
        S = embeds.shape[0]
        if self._gemv_rows_ok(S):
            return self._forward_rows_gemv(embeds, batch, pos0)
        if self._few_rows_ok(S):
            return self._forward_rows_fused(embeds, batch, pos0)
        if self._prompt_fused_ok(S):
            return self._forward_rows_prompt(embeds, batch, pos0)
        x = embeds.to(torch.bfloat16).contiguous().view(1, S, self.H)
        kv_len = pos0 + S
        q = torch.empty(1, S, self.Hq, self.D, dtype=torch.bfloat16, device=self.device)
        o = torch.empty_like(q)
        for l in self.layers:
            kc, vc = l.k_cache[batch:batch + 1], l.v_cache[batch:batch + 1]
            xn = raw.rmsnorm(x, l.norm1, self.eps)
            qkv = raw.linear(xn, self._w(l, "w_qkv"), None, None, 0, False)
            raw.prefill_rope_kv(qkv, q, kc, vc, self.cos, self.sin, pos0, self.Hq, self.Hkv, self.D)
            raw.attention_cache(q, kc, vc, o, kv_len, self.scale)
            x = raw.linear(o.view(1, S, self.Hq * self.D), self._w(l, "w_o"), None, x, 0, False)
            xn = raw.rmsnorm(x, l.norm2, self.eps)
            x = raw.linear(self._mlp_up(xn, self._w(l, "w_gate_up")), self._w(l, "w_down"), None, x, 0, False)
        return x

    # A prompt of up to this many rows (one event prompt is 636) is a GEMM of only 5 x 32 tiles of
    # 128 x 128 on 256 CUs, so its row-parallel projections split K anyway; they stop at the fp32
    # slabs and the kernel that sums them also adds the residual and applies the next RMSNorm, and
    # the q | k | v projection rotates and appends to the cache in its own epilogue: 5 launches a
    # layer instead of 8.
    prompt_fused_rows = 1024

    def _prompt_fused_ok(self, rows: int) -> bool:
        return (32 < rows <= self.prompt_fused_rows and self.H % 64 == 0 and self.I % 64 == 0
                and (self.Hq * self.D) % 64 == 0)

    def _forward_rows_prompt(self, embeds: torch.Tensor, batch: int, pos0: torch.Tensor) -> torch.Tensor:
        raw = self.raw
        S = embeds.shape[0]
        x = embeds.to(torch.bfloat16).contiguous().view(S, self.H)
        kv_len = pos0 + S
        q = torch.empty(1, S, self.Hq, self.D, dtype=torch.bfloat16, device=self.device)
        o = torch.empty_like(q)
        xn = raw.rmsnorm(x, self.layers[0].norm1, self.eps)
        n = len(self.layers)
        rope_gemm = self.D == 128
        for i, l in enumerate(self.layers):
            kc, vc = l.k_cache[batch:batch + 1], l.v_cache[batch:batch + 1]
            w_qkv = self._w(l, "w_qkv")
            if not (rope_gemm and raw.linear_qkv_rope(xn, w_qkv, q, kc, vc, self.cos, self.sin, pos0,
                                                      self.Hq, self.Hkv)):
                rope_gemm = False
                qkv = raw.linear(xn, w_qkv, None, None, 0, False).view(1, S, -1)
                raw.prefill_rope_kv(qkv, q, kc, vc, self.cos, self.sin, pos0, self.Hq, self.Hkv, self.D)
            raw.attention_cache(q, kc, vc, o, kv_len, self.scale)
            x, xn = self._reduce_residual_norm(self._row_partial(o.view(S, self.Hq * self.D), self._w(l, "w_o")),
                                               x, l.norm2)
            h = self._mlp_up(xn, self._w(l, "w_gate_up"))
            nxt = self.layers[i + 1].norm1 if i + 1 < n else None
            x, xn = self._reduce_residual_norm(self._row_partial(h, self._w(l, "w_down")), x, nxt)
        return x.view(1, S, self.H)

    # ----------------------------------------------------- speculative verify
    @torch.no_grad()
    def verify(self, draft: torch.Tensor, n_draft: Optional[int] = None,
               return_hidden: bool = False):
        """Verify ``draft`` (device int32 ``[G]``) against this model in ONE forward.

        The G+1 tokens ``[current, d1..dG]`` are embedded on the device, run through
        the layers at cache positions ``pos..pos+G``, and the lm_head GEMM leaves
        only per-row argmax partials; the fused accept kernel then finds the agreed
        prefix, appends accepted + correction/bonus tokens to the output ring,
        advances ``pos`` (KV rollback = the counter; rejected slots are simply
        overwritten later), flags EOS and stages the next embedding.  Nothing is
        read back: ``self.sd_result`` holds ``[n_accepted, next_token, n_emitted]`` on
        the device.  Returns the G+1 hidden rows (post final-norm) if asked — the
        verifier-side states that B1 / L5F adapters consume.
        """
        G = int(draft.shape[0]) if n_draft is None else int(n_draft)
        if G < 1 or G > 63:
            raise ValueError(f"verify takes 1..63 draft tokens, got {G}")
        if not self.cache_room(G + 1):
            raise ValueError(f"verify of {G} draft tokens at position {self._pos_hi} does not fit the cache "
                             f"({self.l_max}); fall back to decode_step or stop")
        self._pos_hi += G + 1
        draft = draft[:G].to(torch.int32)
        if not self.use_graph:
            hn = self._verify_body(draft.contiguous(), G)
            return hn if return_hidden else None
        # one captured graph per draft length: the whole verify (embed gather, layers,
        # lm_head argmax, accept kernel) replays as a single launch; positions, the
        # current token and the EOS id are all read from device memory
        entry = self._verify_graphs.get((G, self.eos))
        if entry is None:
            with capture_tensor_mode():
                static_draft = torch.zeros(G, dtype=torch.int32, device=self.device)
            static_draft.copy_(draft)
            box = {}
            keep = [t.clone() for t in (self.out_tokens, self.sd_result)]
            g = self._capture(lambda: box.__setitem__("hn", self._verify_body(static_draft, G)))
            self.out_tokens.copy_(keep[0]); self.sd_result.copy_(keep[1])
            entry = (g, static_draft, box["hn"])
            self._verify_graphs[(G, self.eos)] = entry
        g, static_draft, hn = entry
        static_draft.copy_(draft)
        g.replay()
        return hn.clone() if return_hidden else None

    def _verify_body(self, draft: torch.Tensor, G: int) -> torch.Tensor:
        raw = self.raw
        toks = torch.cat([self.cur_token[:1], draft]).long()
        emb = self.embed.index_select(0, toks.clamp_(0, self.embed.shape[0] - 1))
        h = self._forward_rows(emb, 0, self.pos[:1])
        hn = raw.rmsnorm(h, self.final_norm, self.eps)
        pv, pi = raw.linear_argmax_partials(hn.view(G + 1, self.H), self.lm_head, None)
        raw.verify_accept(pv, pi, draft, G, self.out_tokens, self.step, self.pos, self.done,
                          self.cur_token, self.sd_result, self.eos, self.embed, self.x,
                          self.sd_limit, self.sd_stats)
        return hn.view(G + 1, self.H)

    # ------------------------------------------------------------ decode step
    def _layers_step(self) -> None:
        raw = self.raw
        if self.B > int(raw.gemv_max_batch()):
            raise RuntimeError(f"decode steps of more than {int(raw.gemv_max_batch())} sequences run on "
                               "parallel.tp.TensorParallelEngine (also with one rank)")
        for l in self.layers:
            self._gemv_qkv(l, self.x, self.q, l.k_cache, l.v_cache, self.pos, False)
            raw.decode_attn(self.q, l.k_cache, l.v_cache, self.attn, self.part_o, self.part_ml,
                            self.pos, self.nsplit, self.scale)
            self._gemv_residual(l, "w_o", self.attn, self.x)
            self._gemv_mlp_up(l, self.x, self.h)
            self._gemv_residual(l, "w_down", self.h, self.x)

    def _logits_and_commit(self, greedy: bool, want_logits: bool = False) -> None:
        raw = self.raw
        raw.decode_logits(self.lm_head, self.x, self.final_norm, self.eps,
                          None if (greedy and not want_logits) else self.logits,
                          self.part_val, self.part_idx)
        if greedy:
            raw.commit_token(self.part_val, self.part_idx, self.n_part, None, self.embed, self.x,
                             self.out_tokens, self.step, self.pos, self.done, self.cur_token,
                             self.eos, self.active)

    def _commit_sampled(self) -> None:
        self.raw.commit_token(self.part_val, self.part_idx, self.n_part, self.sampled, self.embed,
                              self.x, self.out_tokens, self.step, self.pos, self.done,
                              self.cur_token, self.eos, self.active)

    def _poll_hook(self) -> None:
        """Called where the host has just waited for the device anyway (EOS polls); a
        tensor-parallel engine checks its collectives' error words here."""

    def _step_greedy(self) -> None:
        self._layers_step()
        self._logits_and_commit(greedy=True)

    def _step_logits(self) -> None:
        self._layers_step()
        self._logits_and_commit(greedy=False)

    def _capture(self, fn):
        """Capture ``fn`` into a HIP graph.  Warm-up and capture advance the device
        counters, so they are snapshotted and restored (also when the capture fails)."""
        state = (self.step, self.pos, self.done, self.cur_token, self.x)
        saved = [t.clone() for t in state]
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()  # warm-up (first-launch module load) outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: another host thread driving a second engine on
            # its own stream must not invalidate this capture
            with capture_tensor_mode(), no_gc_during_capture(), \
                    torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
        finally:
            for t, s in zip(state, saved):
                t.copy_(s)
            torch.cuda.synchronize()
        return g

    def _capture_or_eager(self, fn):
        """``_capture`` that degrades to eager stepping when the step cannot be captured
        (e.g. a collective backend that does not record into graphs): warns once, turns
        graphs off for this engine, records the reason in ``graph_fallback`` (benchmarks
        report it, so an eager run cannot pass for a captured one) and returns None."""
        try:
            return self._capture(fn)
        except RuntimeError as e:
            import warnings
            warnings.warn(f"decode step could not be captured into a HIP graph ({e}); running eagerly")
            self.use_graph = False
            self.graph_fallback = str(e)[:200]
            return None

    graph_fallback: Optional[str] = None

    @property
    def graph_captured(self) -> bool:
        """True when decode steps replay a captured graph (not the eager fallback)."""
        return bool(self.use_graph and (self._graph_greedy or self._graph_logits is not None))

    def ensure_graphs(self, eos_token_id: Optional[int] = None, greedy: bool = True) -> None:
        """Capture the decode graph for this EOS id now, if it is not captured yet.
        Schedulers call this before they start timed or multi-threaded regions, so
        that no capture (which synchronises the device) lands inside them."""
        if eos_token_id is not None:
            self.eos = int(eos_token_id)
        if not self.use_graph:
            return
        if greedy:
            if self.eos not in self._graph_greedy:
                self._graph_greedy[self.eos] = self._capture(self._step_greedy)
        elif self._graph_logits is None:
            self._graph_logits = self._capture(self._step_logits)

    def decode_step(self, greedy: bool = True) -> None:
        """Advance every sequence by one token (graph replay when enabled)."""
        self._pos_hi += 1
        if not self.use_graph:
            (self._step_greedy if greedy else self._step_logits)()
            return
        if greedy:
            g = self._graph_greedy.get(self.eos)
            if g is None:
                g = self._capture_or_eager(self._step_greedy)
                if g is None:
                    return self._step_greedy()
                self._graph_greedy[self.eos] = g
            g.replay()
        else:
            if self._graph_logits is None:
                self._graph_logits = self._capture_or_eager(self._step_logits)
                if self._graph_logits is None:
                    return self._step_logits()
            self._graph_logits.replay()

    # --------------------------------------------------------------- generate
    def _sample_and_commit(self, temperature: float, top_p: float, generator, top_k: int = 0) -> None:
        """Draw the next token on the device (top-k / top-p by bisection, keyed by
        ``(seed, step)``) and commit it; no host round trip."""
        seed = int(generator.initial_seed()) if generator is not None else 0
        self.raw.sample_top_p(self.logits, self.sampled, float(temperature), float(top_p),
                              seed & 0x7FFFFFFFFFFFFFFF, self.step, int(top_k))
        self._commit_sampled()

    @torch.no_grad()
    def generate(self, embeds: torch.Tensor, max_new_tokens: int = 512, temperature: float = 0.0,
                 top_p: float = 1.0, eos_token_id: Optional[int] = None,
                 generator: Optional[torch.Generator] = None, top_k: int = 0) -> torch.Tensor:
        """Greedy / sampled generation for one sequence; returns new token ids (long).
        Sampling applies temperature, then ``top_k`` (0 = off), then ``top_p``.

        Greedy decoding replays one captured graph per token and syncs with the
        host only every ``poll_every`` tokens to read the EOS flag.  Sampling
        replays the layers+logits graph, then draws the token on the device.
        """
        if eos_token_id is not None:
            self.eos = int(eos_token_id)
        if max_new_tokens > self.max_out:
            raise ValueError(f"max_new_tokens {max_new_tokens} > engine ring {self.max_out}")
        self.reset()
        greedy = temperature <= 0
        self.prefill(embeds, select="greedy" if greedy else "logits")
        if not greedy:
            self._sample_and_commit(temperature, top_p, generator, top_k)
        budget = min(max_new_tokens, self.l_max - embeds.shape[0])
        return self.generate_continue(budget, None, temperature, top_p, generator, top_k)

    @torch.no_grad()
    def score(self, embeds: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """Teacher-forced logits of sequence 0: row 0 predicts the token after the prompt, row
        ``i`` the token after ``tokens[i-1]`` -- ``[len(tokens)+1, V]`` fp32.  The given tokens
        are committed instead of the model's own choice (one decode step each), so two
        implementations can be compared on logits along the SAME path rather than on how
        long their greedy outputs happen to agree."""
        eos, self.eos = self.eos, -1
        self.reset()
        self.prefill(embeds, select="logits")
        rows = [self.logits[0].float().clone()]
        for t in tokens.tolist():
            self.sampled[0] = int(t)
            self._commit_sampled()
            self.decode_step(greedy=False)
            rows.append(self.logits[0].float().clone())
        self.eos = eos
        return torch.stack(rows)

    @torch.no_grad()
    def generate_continue(self, max_new_tokens: int, eos_token_id: Optional[int] = None,
                          temperature: float = 0.0, top_p: float = 1.0,
                          generator: Optional[torch.Generator] = None, top_k: int = 0) -> torch.Tensor:
        """Keep decoding sequence 0 until ``max_new_tokens`` tokens have been emitted in
        total (counting those already committed) or EOS; returns all emitted tokens."""
        if eos_token_id is not None:
            self.eos = int(eos_token_id)
        greedy = temperature <= 0
        n = int(self.step[0].item())
        budget = min(max_new_tokens, self.max_out)
        while n < budget and not int(self.done[0].item()):
            self._poll_hook()
            chunk = min(self.poll_every, budget - n)
            for _ in range(chunk):
                self.decode_step(greedy)
                if not greedy:
                    self._sample_and_commit(temperature, top_p, generator, top_k)
            n += chunk
        toks = self.out_tokens[0, :n].to(torch.long)
        if self.eos >= 0:
            hit = (toks == self.eos).nonzero()
            if hit.numel():
                toks = toks[: int(hit[0]) + 1]
        return toks
