"""Differentiable GPU path: the ops a training step differentiates through, in the
tensors' own dtype.

``ops.reference`` is the fp32 oracle (it upcasts every operand, which for a 7B decoder
means re-materialising each weight matrix in fp32 on every call).  A call on GPU tensors
that needs autograd comes here instead.  On bf16 tensors every op is an
``autograd.Function`` over the in-tree gfx950 kernels:

* ``linear``: forward, dgrad ``dX = dY W`` and wgrad ``dW = dY^T X`` all run the MFMA GEMM
  of ``csrc/kernels/gemm.hip``; the backward products read their K-strided operands
  (``W`` for dgrad, both ``dY`` and ``X`` for wgrad) through transposing LDS reads, so no
  transposed copy of a weight or an activation is ever written;
* ``attention``: the fused forward also returns the per-row log-sum-exp, and the backward
  (``csrc/kernels/attn_bwd.hip``) recomputes the probabilities from it;
* RMSNorm, RoPE and SwiGLU: the inference kernel in forward and one backward kernel each.

Other dtypes (fp32 parameters under autocast, shapes the kernels do not take) fall back to
PyTorch-ROCm library ops with the same semantics; rounding is that of bf16 training.
The reference trains through HF / PEFT modules on cuBLAS and flash-attn
(pipeline/adapter_train/train_lora_adapter.py:458-524).

RMSNorm, RoPE and SwiGLU on bf16 tensors are ``autograd.Function``s over the HIP kernels:
the inference kernel in forward and one backward kernel each (``csrc/kernels/train_bwd.hip``;
RoPE's backward is the forward kernel with the angle negated).  Written as eager tensor
expressions they are 8-12 launches per call, ~2 000 per 7B forward, and the step is bound by
the host's launch rate (87 ms forward for ~840 tokens against 13 ms of kernel time).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .reference import _ACTS


def _fused(*tensors) -> bool:
    """bf16 device tensors and the HIP extension at hand: take the fused Functions."""
    return all(t is None or (t.is_cuda and t.dtype == torch.bfloat16) for t in tensors) and _raw() is not None


_RAW = []


def _raw():
    if not _RAW:
        try:
            from . import hip
            _RAW.append(hip.raw)
        except Exception:          # extension not built: eager expressions below
            _RAW.append(None)
    return _RAW[0]


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        xc = x.contiguous()
        ctx.save_for_backward(xc, weight)
        ctx.eps = eps
        return _raw().rmsnorm(xc, weight.contiguous(), eps)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        want_dw = bool(ctx.needs_input_grad[1])
        dx, dw = _raw().rmsnorm_bwd(x, w.contiguous(), g.contiguous(), ctx.eps, want_dw)
        return dx, (dw.to(w.dtype).view_as(w) if want_dw else None), None


class _Rope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, positions, cos, sin):
        pos = positions.to(torch.long).contiguous()
        ctx.save_for_backward(pos, cos, sin)
        return _raw().rope(x.contiguous(), pos, cos, sin)

    @staticmethod
    def backward(ctx, g):
        pos, cos, sin = ctx.saved_tensors
        return _raw().rope(g.contiguous(), pos, cos, -sin), None, None, None   # rotate back by the same angle


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate_up):
        gu = gate_up.contiguous()
        ctx.save_for_backward(gu)
        return _raw().swiglu(gu)

    @staticmethod
    def backward(ctx, g):
        (gu,) = ctx.saved_tensors
        return _raw().swiglu_bwd(gu, g.contiguous())


def rmsnorm(x, weight, eps: float = 1e-6):
    if _fused(x, weight) and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192:
        return _RMSNorm.apply(x, weight, eps)
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)
    return y * weight.to(x.dtype)


def add_rmsnorm(x, residual, weight, eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    r = x + residual
    return rmsnorm(r, weight, eps), r


class _LayerNorm(torch.autograd.Function):
    """LayerNorm on bf16 activations: the inference kernel forward, one backward kernel
    (``csrc/kernels/adapter_train.hip``).  Weight and bias may be fp32 parameters (training under bf16
    autocast): they are rounded to bf16 for the kernels and receive fp32 gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xc = x.contiguous()
        wb = weight.detach().to(torch.bfloat16).contiguous()
        bb = None if bias is None else bias.detach().to(torch.bfloat16).contiguous()
        ctx.save_for_backward(xc, wb)
        ctx.eps, ctx.has_bias = eps, bias is not None
        ctx.pdtype = (weight.dtype, None if bias is None else bias.dtype)
        return _raw().layernorm(xc, wb, bb, eps, 0)

    @staticmethod
    def backward(ctx, g):
        x, wb = ctx.saved_tensors
        want = bool(ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]))
        dx, dw, db = _raw().layernorm_bwd(x, wb, g.contiguous(), ctx.eps, want)
        return (dx, dw.to(ctx.pdtype[0]).view_as(wb) if ctx.needs_input_grad[1] else None,
                db.to(ctx.pdtype[1]).view_as(wb) if (ctx.has_bias and ctx.needs_input_grad[2]) else None, None)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        return _raw().gelu_fwd(xc)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return _raw().gelu_bwd(x, g.contiguous())


class _Dropout(torch.autograd.Function):
    """Inverted dropout whose keep mask is a counter-based hash of (seed, element index): nothing is
    stored for the backward, which re-applies the same mask to the gradient."""

    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return _raw().dropout_fwd(x.contiguous(), p, seed)

    @staticmethod
    def backward(ctx, g):
        return _raw().dropout_fwd(g.contiguous(), ctx.p, ctx.seed), None, None


def next_seed() -> int:
    """A fresh seed for the counter-based dropout masks, drawn from torch's CPU generator (so
    ``torch.manual_seed`` makes a run repeatable)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def _ew_ok(x) -> bool:
    return _fused(x) and x.numel() > 0 and x.numel() % 8 == 0


def layernorm(x, weight, bias, eps: float = 1e-5):
    h = x.shape[-1]
    if (_fused(x) and weight is not None and weight.is_cuda and weight.numel() == h and h % 8 == 0 and h <= 8192
            and x.numel() > 0 and (bias is None or bias.is_cuda)):
        return _LayerNorm.apply(x, weight, bias, eps)
    return F.layer_norm(x, (h,), None if weight is None else weight.to(x.dtype),
                        None if bias is None else bias.to(x.dtype), eps)


def gelu(x):
    return _Gelu.apply(x) if _ew_ok(x) else F.gelu(x)


def dropout(x, p: float, training: bool = True):
    if not training or p <= 0.0:
        return x
    if _ew_ok(x) and p < 1.0:
        return _Dropout.apply(x, float(p), next_seed())
    return F.dropout(x, p, True)


class _AttnD512(torch.autograd.Function):
    """Self-attention with head_dim 512 over a packed in-projection ``[B, S <= 64, 3 * H * 512]``
    (``csrc/kernels/adapter_train.hip``): the forward keeps the 64 x 64 probabilities of every
    head, the backward returns the gradient of the packed projection in one tensor."""

    @staticmethod
    def forward(ctx, qkv, heads, causal, kpm, scale, drop_p, seed):
        qc = qkv.contiguous()
        out, probs = _raw().attn512_fwd(qc, heads, bool(causal), kpm, float(scale), float(drop_p), int(seed), True)
        ctx.save_for_backward(qc, probs)
        ctx.args = (heads, float(scale), float(drop_p), int(seed))
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, probs = ctx.saved_tensors
        heads, scale, drop_p, seed = ctx.args
        return _raw().attn512_bwd(qkv, g.contiguous(), probs, heads, scale, drop_p, seed), None, None, None, None, None, None


def attention_d512_ok(qkv, heads: int) -> bool:
    return (_fused(qkv) and qkv.dim() == 3 and qkv.shape[-1] == 3 * heads * 512 and 1 <= qkv.shape[1] <= 64
            and qkv.shape[0] >= 1)


def attention_d512(qkv, heads: int, causal: bool = False, key_padding_mask=None, dropout_p: float = 0.0):
    """``[B, S, 3 * heads * 512]`` packed q | k | v (``nn.MultiheadAttention``'s in-projection layout)
    -> ``[B, S, heads * 512]``.  ``key_padding_mask``: bool ``[B, S]``, True = padded key."""
    kpm = None if key_padding_mask is None else key_padding_mask.to(torch.bool).contiguous()
    seed = next_seed() if dropout_p > 0 else 0
    return _AttnD512.apply(qkv, heads, causal, kpm, 1.0 / math.sqrt(512), dropout_p, seed)


def _round64(n: int) -> int:
    return (n + 63) // 64 * 64


def _pad_cols(t: torch.Tensor, k: int) -> torch.Tensor:
    return t if t.shape[1] == k else F.pad(t, (0, k - t.shape[1]))


def linear_dgrad(gy: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``dX [M,K] = dY [M,N] @ W [N,K]``: the contraction runs over W's ROWS, so W is the
    K-strided operand of the GEMM (read through transposing LDS loads, never copied)."""
    n = gy.shape[1]
    if n % 8:                 # 16-byte chunks; otherwise the kernel itself stops at dY's last column
        gy = _pad_cols(gy, _round64(n))
    return _raw().linear_t(gy, weight, False, True, None, None, 0, False, _round64(n))


def linear_wgrad(gy: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``dW [N,K] = dY^T [N,M] @ X [M,K]``: both operands are strided in the token dimension.
    ``out``: a contiguous bf16 ``[N,K]`` tensor the product is written into (an optimizer's
    flat gradient buffer), instead of a new tensor."""
    # the contraction runs to the next multiple of 64; both operands are K-strided, whose staging
    # loads return zeros for the rows past the last token, so neither is padded
    return _raw().linear_t(gy, x, True, True, None, None, 0, False, _round64(gy.shape[0]), out)


def enable_transposed_dgrad(module: torch.nn.Module, min_free_bytes: int = 32 << 30, group: bool = True) -> int:
    """Give every FROZEN bf16 weight matrix of ``module`` a row-major copy of its transpose for the
    input-gradient product.  ``dX = dY W`` contracts over W's rows; read from ``W`` itself that is the
    K-strided operand path (transposing LDS reads, 420-660 TFLOP/s at ~840 tokens), read from
    ``W^T [K, N]`` it is the plain product the forward runs (750-880).  Only weights that do not
    train qualify (the copy would be stale after every optimizer step), and only while at least
    ``min_free_bytes`` of device memory stay free.  Returns the bytes spent.  A copy is refreshed
    if its weight was written in place or moved since (``_version`` / ``data_ptr``).
    ``group``: first let every submodule that can (``group_frozen``: gate / up of an MLP) gather its
    frozen projections in one matrix; the group then gets ONE transposed copy."""
    spent = 0
    if group:
        for m in module.modules():
            if hasattr(m, "group_frozen"):
                m.group_frozen()
    todo = []
    for m in module.modules():
        todo += [g for g in (m.__dict__.get(a) for a in _GROUP_ATTRS) if g is not None]
        w = getattr(m, "weight", None)
        if isinstance(m, torch.nn.Linear) and w is not None and not getattr(w, "_in_group", False):
            todo.append(w)
    for w in todo:
        if w.requires_grad or not _fused(w):
            continue
        n, k = w.shape
        if n % 64 or k % 8 or getattr(w, "_dgrad_t", None) is not None:
            continue
        need = w.numel() * w.element_size()
        if torch.cuda.mem_get_info(w.device)[0] - need < min_free_bytes:
            break
        w._dgrad_t = (w.detach().t().contiguous(), _version_of(w), w.data_ptr())
        spent += need
    return spent


# module attributes that hold a group matrix (``models.llama.group_frozen_weights``)
_GROUP_ATTRS = ("_w_gate_up", "_w_qkv")


def _version_of(w: torch.Tensor) -> int:
    """In-place writes so far, through the tensor itself or -- for a group matrix -- through the
    member weights that are views of it (a view set with ``.data`` counts its own writes)."""
    return w._version + sum(m._version for m in getattr(w, "_members", ()))


def release_transposed_dgrad(module: torch.nn.Module, ungroup: bool = True) -> int:
    """Drop the copies ``enable_transposed_dgrad`` made (after training: they are one more copy of
    every frozen matrix) and, with ``ungroup``, dissolve the weight groups it formed.  Returns the
    bytes the copies held."""
    freed = 0
    groups = [g for m in module.modules() for g in (m.__dict__.get(a) for a in _GROUP_ATTRS) if g is not None]
    for p in list(module.parameters()) + groups:
        ent = getattr(p, "_dgrad_t", None)
        if ent is not None:
            freed += ent[0].numel() * ent[0].element_size()
            del p._dgrad_t
    if ungroup:
        # the grouped projections get their own storage back (a module moved to another device later
        # would otherwise leave the group matrix behind on this one)
        for m in module.modules():
            for a in _GROUP_ATTRS:
                cat = m.__dict__.pop(a, None)
                if cat is None:
                    continue
                lo, hi = cat.data_ptr(), cat.data_ptr() + cat.numel() * cat.element_size()
                for w in getattr(cat, "_members", ()):
                    if lo <= w.data_ptr() < hi:
                        w.data = w.data.clone()
                    w._in_group = False
    return freed


def _transposed_for_dgrad(weight) -> Optional[torch.Tensor]:
    ent = getattr(weight, "_dgrad_t", None)
    if ent is None or weight.requires_grad:
        return None
    wt, version, ptr = ent
    if version != _version_of(weight) or ptr != weight.data_ptr():
        wt.copy_(weight.detach().t())
        weight._dgrad_t = (wt, _version_of(weight), weight.data_ptr())
    return wt


def _wgrad_to_sink(weight, g2, x2):
    """``dW = g2^T x2``, written into the parameter's published gradient slot if that is still free."""
    sink = getattr(weight, "_grad_sink", None)
    if (sink is not None and weight.grad is None and not getattr(weight, "_grad_sink_used", True)
            and sink.dtype == g2.dtype and sink.shape == weight.shape and sink.is_contiguous()):
        weight._grad_sink_used = True
        return linear_wgrad(g2, x2, out=sink).view_as(weight)
    return linear_wgrad(g2, x2)


class _Linear(torch.autograd.Function):
    """``y = x @ W^T + b`` on the in-tree MFMA GEMM, forward and both backward products."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None):
        k = x.shape[-1]
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(xc.view(-1, k), weight)
        ctx.has_bias = bias is not None
        from . import hip
        # the kernel's own output in the input's leading shape: not a view, so a caller may update it in place;
        # a residual is added by the GEMM's epilogue (its gradient is the incoming gradient itself)
        return hip.linear(xc, weight, bias, None, residual)

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        n = weight.shape[0]
        g2 = gy.reshape(-1, n)
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _transposed_for_dgrad(weight)
            if wt is not None:
                from . import hip
                dx = hip.linear(g2, wt).view(*gy.shape[:-1], weight.shape[1])
            else:
                dx = linear_dgrad(g2, weight.contiguous()).view(*gy.shape[:-1], weight.shape[1])
        if ctx.needs_input_grad[1]:
            # an optimizer that keeps gradients in a flat buffer publishes the parameter's slot
            # as ``_grad_sink``: the first gradient of a step is written there by the GEMM itself
            # and handed to autograd as that view, which it adopts as ``.grad`` without a copy
            # (later gradients of the same step are accumulated by autograd as usual)
            # A weight used several times in one graph gets all its gradients summed by autograd
            # BEFORE ``.grad`` is set, so "the slot is free" is tracked by the optimizer's own flag
            # (cleared by its ``zero_grad``), never inferred from ``.grad`` alone.
            dw = _wgrad_to_sink(weight, g2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g2.float().sum(0).to(gy.dtype)
        return dx, dw, db, (gy if ctx.needs_input_grad[3] else None)


def _linear_fused_ok(x, weight, bias) -> bool:
    if not _fused(x, weight, bias) or weight.dim() != 2 or x.numel() == 0:
        return False
    n, k = weight.shape
    # K-strided operands move in 16-byte chunks along their contiguous dimension
    return n % 8 == 0 and k % 8 == 0 and n >= 8 and k >= 8 and x.shape[-1] == k


def linear(x, weight, bias=None, act: Optional[str] = None, residual=None):
    if _linear_fused_ok(x, weight, bias):
        if residual is not None and not act and _fused(residual) and residual.shape[:-1] == x.shape[:-1] \
                and residual.shape[-1] == weight.shape[0]:
            return _Linear.apply(x, weight, bias, residual)
        y = _Linear.apply(x, weight, bias, None)
    else:
        y = F.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))
    y = _ACTS[act](y)
    return y if residual is None else y + residual


class _GatherRows(torch.autograd.Function):
    """``x[idx]`` for ascending unique row indices; the backward writes the whole ``[R, K]`` gradient in
    one launch (the gathered rows, zeros elsewhere) instead of a zero fill plus an indexed accumulation."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.rows = x.shape[0]
        return _raw().rows_gather(x.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _raw().rows_scatter(g.contiguous(), idx, ctx.rows), None


class _SpliceRows(torch.autograd.Function):
    """Input embeddings of a multimodal prompt from a FROZEN token table and event features that train:
    row (b, l) is ``table[src]`` for ``src >= 0``, ``feats[-1 - src]`` for an event slot, zeros for padding
    (one launch, ``splice_rows``).  Every feature row is placed exactly once, so the features' gradient is a
    row gather through the inverse map ``inv[j]`` = flat output position of feature row ``j`` -- not the
    sort-and-accumulate an indexed read's backward runs."""

    @staticmethod
    def forward(ctx, table, feats, src, inv):
        ctx.save_for_backward(inv)
        return _raw().splice_rows(table, feats.contiguous(), src)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        return None, _raw().rows_gather(g2, inv), None, None


def splice_rows(table, feats, src, inv):
    """See ``_SpliceRows``; ``src`` int32 ``[B, L]``, ``inv`` int32 ``[feats.shape[0]]`` (both on the device)."""
    return _SpliceRows.apply(table, feats, src, inv)


class _LinearCrossEntropy(torch.autograd.Function):
    """Mean token cross-entropy of ``x W^T`` against ``labels`` (``csrc/kernels/loss.hip``): bf16 logits
    from the in-tree GEMM, log-sum-exp and loss per row in one pass, and in the backward the logits'
    gradient written once, in bf16, into a row-padded buffer that the dgrad / wgrad GEMMs read.  No fp32
    copy of the logits, no softmax tensor, and the upstream gradient never visits the host."""

    @staticmethod
    def forward(ctx, x, weight, labels, ignore, row_weight):
        from . import hip
        logits = hip.linear(x, weight)                           # [M, V] bf16
        lab = labels.to(torch.int32).contiguous()
        lse, _, stat = _raw().ce_fwd(logits, lab, ignore, row_weight)
        ctx.save_for_backward(x, weight, logits, lab, lse, stat, row_weight)
        ctx.ignore = ignore
        return stat[0].clone()

    @staticmethod
    def backward(ctx, g):
        x, weight, logits, lab, lse, stat, row_weight = ctx.saved_tensors
        v, k = weight.shape
        up = g.reshape(1).to(torch.float32)
        d = _raw().ce_bwd(logits, lab, lse, up, stat, ctx.ignore, row_weight)   # [M, round_up(V, 64)], zero padding
        vp = d.shape[1]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = _transposed_for_dgrad(weight)
            if wt is not None and v % 8 == 0:
                from . import hip
                dx = hip.linear(d[:, :v].contiguous() if vp != v else d, wt)
            else:
                # contraction over the vocabulary: W's rows; rows past V read as zeros, d's padding is zero
                dx = _raw().linear_t(d, weight.contiguous(), False, True, None, None, 0, False, vp)
        if ctx.needs_input_grad[1]:
            full = _raw().linear_t(d, x, True, True, None, None, 0, False, _round64(x.shape[0]))   # [Vp, K]
            dw = full[:v]
        return dx, dw, None, None, None


def linear_cross_entropy_ok(x, weight) -> bool:
    return (_fused(x, weight) and weight.dim() == 2 and x.shape[-1] == weight.shape[1] and weight.shape[1] % 8 == 0
            and weight.shape[0] >= 8 and weight.shape[0] * weight.shape[1] < (1 << 30))


def linear_cross_entropy(hidden, weight, labels, ignore_index: int = -100, host_labels=None,
                         per_sequence_mean: bool = False):
    """``cross_entropy(hidden @ weight^T, labels)`` (mean over the labels that are not ``ignore_index``)
    with the projection restricted to the supervised rows.  ``hidden [.., K]``, ``labels [..]``;
    ``host_labels``: a CPU copy of ``labels`` if the caller has one (the row selection is host
    arithmetic; without it the labels are read back once).  ``per_sequence_mean`` (``labels [B, S]``):
    the mean over sequences of each sequence's own token mean -- what B separate single-sequence
    steps with their losses averaged compute -- instead of the mean over all supervised tokens."""
    k = hidden.shape[-1]
    x = hidden.reshape(-1, k)
    lab = labels.reshape(-1)
    lab_h2 = host_labels if host_labels is not None else labels.detach().cpu()
    lab_h = lab_h2.reshape(-1)
    live = lab_h != ignore_index
    rows = live.nonzero().reshape(-1)
    wrow = None
    if per_sequence_mean and lab_h2.dim() == 2 and rows.numel():
        per = (lab_h2 != ignore_index).sum(1)                        # supervised tokens of each sequence
        n_seq = int((per > 0).sum())
        wfull = (1.0 / (per.clamp(min=1).float() * max(n_seq, 1)))[:, None].expand_as(lab_h2).reshape(-1)
        wrow = wfull[rows].contiguous()
    if not linear_cross_entropy_ok(x, weight) or k % 8 or rows.numel() == 0:
        if not rows.numel():
            return x.sum() * 0.0
        sel = rows.to(x.device)
        per_row = F.cross_entropy(F.linear(x[sel], weight.to(x.dtype)).float(), lab[sel], reduction="none")
        return per_row.mean() if wrow is None else (per_row * wrow.to(x.device)).sum()
    idx = rows.to(torch.int32).to(x.device, non_blocking=True)
    xs = _GatherRows.apply(x, idx) if rows.numel() < x.shape[0] else x.contiguous()
    ls = lab_h[rows].to(torch.int32).to(x.device, non_blocking=True)
    wd = None if wrow is None else wrow.to(x.device, non_blocking=True)
    return _LinearCrossEntropy.apply(xs, weight, ls, int(ignore_index), wd)


class _LowRank(torch.autograd.Function):
    """``(scale * x A^T) B^T`` of a low-rank adapter (A ``[r, K]``, B ``[N, r]``, r <= 16).  The two
    products with an r-wide OUTPUT -- ``x A^T`` forward, ``dY B`` backward -- run the rank-r kernel (one
    launch each, the scale applied to its fp32 sums) instead of a tiled GEMM with a K split and its
    reduction pass; the wide ones (``t B^T``, ``dt A`` and both weight gradients) stay on the GEMM."""

    @staticmethod
    def forward(ctx, x, a, b, scale):
        from . import hip
        xc = x if x.is_contiguous() else x.contiguous()
        t = _raw().lowrank_out(xc, a, scale)                       # [.., r]
        ctx.save_for_backward(xc.view(-1, xc.shape[-1]), t.view(-1, t.shape[-1]), a, b)
        ctx.scale = scale
        return hip.linear(t, b)                                    # [.., N], the kernel's own output

    @staticmethod
    def backward(ctx, gy):
        x2, t2, a, b = ctx.saved_tensors
        g2 = gy.reshape(-1, b.shape[0])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        dx = da = db = None
        dt = _raw().lowrank_out(g2, b.detach().t().contiguous(), ctx.scale)      # scale * dY B  [M, r]
        if ctx.needs_input_grad[2]:
            db = _wgrad_to_sink(b, g2, t2)
        if ctx.needs_input_grad[1]:
            da = _wgrad_to_sink(a, dt, x2)
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dt, a.detach().contiguous()).view(*gy.shape[:-1], a.shape[1])
        return dx, da, db, None


def _lora_projection_forward(x, weight, residual, scale, drop_p, offsets, ab):
    """The forward launches of ``_LoRAProjection``: ``(y, x as rows, rank-r intermediates, seeds)``.  One seed per
    adapter is drawn from torch's CPU generator, in adapter order, whether or not autograd records the call --
    so a block that is recomputed under activation checkpointing (which restores that generator) sees the
    masks of its first pass."""
    from . import hip
    raw = _raw()
    k = x.shape[-1]
    xc = x if x.is_contiguous() else x.contiguous()
    x2 = xc.view(-1, k)
    y = hip.linear(xc, weight, None, None, residual)
    y2 = y.view(-1, y.shape[-1])
    ts, seeds = [], []
    for i, off in enumerate(offsets):
        a, b = ab[2 * i], ab[2 * i + 1]
        seed = next_seed() if drop_p > 0 else 0
        t = raw.lowrank_out(x2, a, scale, drop_p, seed)                  # [M, r], scale / (1 - p) applied
        raw.linear_rows_(t, b, y2[:, off:off + b.shape[0]], True)
        ts.append(t)
        seeds.append(seed)
    return y, x2, ts, seeds


class _LoRAProjection(torch.autograd.Function):
    """``y = x W^T (+ residual)`` of a FROZEN projection with low-rank adapters on column ranges of its output
    (one adapter on all columns of an ``o_proj``; three on the q / k / v ranges of a grouped projection):
    ``y[:, off_i : off_i + n_i] += (s / (1 - p)) (m_i (.) x) A_i^T B_i^T``, ``m_i`` the adapter's input-dropout
    keep mask.  One node instead of a dropout, a rank-r node, a strided add and their backward copies per
    adapter:

    * forward: base GEMM (residual in its epilogue), then per adapter the rank-r kernel with the mask applied
      while ``x`` is loaded and a GEMM that accumulates ``t_i B_i^T`` in place onto its column block;
    * backward: the base input gradient, then per adapter ``dt_i`` and ``dB_i`` read the incoming gradient's
      column block where it is (row-strided operands), ``dA_i`` contracts with ``m_i (.) x`` recomputed from
      the seed, and ``m_i (.) (dt_i A_i)`` is accumulated onto the input gradient in place.

    Saved per node: ``x`` once and the ``[rows, r]`` intermediates -- not one dropped copy of ``x`` per adapter."""

    @staticmethod
    def forward(ctx, x, weight, residual, scale, drop_p, offsets, *ab):
        y, x2, ts, seeds = _lora_projection_forward(x, weight, residual, scale, drop_p, offsets, ab)
        ctx.save_for_backward(x2, weight, *ts, *ab)
        ctx.meta = (float(scale), float(drop_p), tuple(offsets), tuple(seeds))
        return y

    @staticmethod
    def backward(ctx, gy):
        raw = _raw()
        scale, p, offsets, seeds = ctx.meta
        n_ad = len(offsets)
        saved = ctx.saved_tensors            # read once: activation checkpointing unpacks on access
        x2, weight = saved[:2]
        ts = saved[2:2 + n_ad]
        ab = saved[2 + n_ad:]
        ntot = weight.shape[0]
        g2 = gy.reshape(-1, ntot)
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        dx2 = None
        if ctx.needs_input_grad[0]:
            wt = _transposed_for_dgrad(weight)
            if wt is not None:
                from . import hip
                dx2 = hip.linear(g2, wt)
            else:
                dx2 = linear_dgrad(g2, weight.contiguous())
        grads = []
        s_eff = scale / (1.0 - p)
        kpad = _round64(g2.shape[0])
        for i, off in enumerate(offsets):
            a, b, t = ab[2 * i], ab[2 * i + 1], ts[i]
            gi = g2[:, off:off + b.shape[0]]                                  # a view: rows strided by ntot
            need_a, need_b = ctx.needs_input_grad[6 + 2 * i], ctx.needs_input_grad[7 + 2 * i]
            da = db = None
            dt = None
            if need_a or dx2 is not None:
                dt = raw.lowrank_out(gi, b.detach().t().contiguous(), s_eff, 0.0, 0)      # s / (1 - p) * dY_i B_i
            if need_b:
                sink = _free_sink(b, gi)
                db = raw.linear_t(gi, t, True, True, None, None, 0, False, kpad, sink)
                db = db.view_as(b)
            if need_a:
                xm = raw.lowrank_mask(x2, p, seeds[i]) if p > 0 else x2
                da = _wgrad_to_sink(a, dt, xm)
            if dx2 is not None:
                if p > 0:
                    raw.lowrank_mask_add_(dx2, linear_dgrad(dt, a.detach().contiguous()), p, seeds[i])
                else:
                    raw.linear_t(dt, a.detach().contiguous(), False, True, None, dx2, 0, False, _round64(a.shape[0]), dx2)
            grads += [da, db]
        dx = None if dx2 is None else dx2.view(*gy.shape[:-1], x2.shape[1])
        return (dx, None, (gy if ctx.needs_input_grad[2] else None), None, None, None, *grads)


def _free_sink(weight, g2):
    """The parameter's published gradient slot if it is still free this step (marks it used), else None."""
    sink = getattr(weight, "_grad_sink", None)
    if (sink is not None and weight.grad is None and not getattr(weight, "_grad_sink_used", True)
            and sink.dtype == g2.dtype and sink.shape == weight.shape and sink.is_contiguous()):
        weight._grad_sink_used = True
        return sink
    return None


def lora_projection_ok(x, weight, adapters, residual=None) -> bool:
    """Shapes and dtypes ``lora_projection`` takes: bf16 GPU tensors, a frozen base, rank <= 16 (a multiple of 8)."""
    if not _fused(x, weight) or weight.requires_grad or weight.dim() != 2 or x.numel() == 0:
        return False
    n, k = weight.shape
    if n % 8 or k % 8 or x.shape[-1] != k or (residual is not None and not _fused(residual)):
        return False
    for off, a, b in adapters:
        if not _fused(a, b) or a.dim() != 2 or b.dim() != 2 or not a.is_contiguous() or not b.is_contiguous():
            return False
        r = a.shape[0]
        if r > 16 or r % 8 or a.shape[1] != k or b.shape[1] != r or b.shape[0] % 8 or off % 8 or off + b.shape[0] > n:
            return False
    return True


def lora_projection(x, weight, adapters, scale: float, drop_p: float = 0.0, residual=None):
    """``x W^T (+ residual)`` with low-rank updates ``adapters = [(column offset, A [r, K], B [n, r]), ...]`` on
    column ranges of the output, input dropout ``drop_p`` per adapter (pass 0 outside training).  Outside
    autograd (evaluation, or the first pass of a checkpointed block) the same kernels run unrecorded."""
    flat = [t for _, a, b in adapters for t in (a, b)]
    offsets = tuple(int(off) for off, _, _ in adapters)
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, residual, *flat))):
        with torch.no_grad():
            return _lora_projection_forward(x, weight, residual, float(scale), float(drop_p), offsets,
                                            [t.detach() for t in flat])[0]
    return _LoRAProjection.apply(x, weight, residual, float(scale), float(drop_p), offsets, *flat)


def lowrank_delta(x, a, b, scale: float):
    """``(scale * x A^T) B^T``: the update a low-rank adapter adds to its layer's output."""
    r, k = a.shape
    if (_fused(x, a, b) and x.numel() > 0 and r <= 16 and r % 8 == 0 and k % 8 == 0 and b.shape[0] % 8 == 0
            and x.shape[-1] == k and b.shape[1] == r and a.is_contiguous() and b.is_contiguous()):
        return _LowRank.apply(x, a, b, float(scale))
    return linear(linear(x, a) * scale, b)


def swiglu(gate_up):
    if _fused(gate_up) and gate_up.shape[-1] % 16 == 0:
        return _SwiGLU.apply(gate_up)
    g, u = gate_up.chunk(2, dim=-1)
    return F.silu(g) * u


def apply_rope(x, positions, cos, sin):
    if (_fused(x) and x.dim() == 4 and x.shape[-1] % 16 == 0 and cos.dtype == torch.float32
            and positions.numel() == x.shape[0] * x.shape[1]):
        return _Rope.apply(x, positions, cos, sin)
    d2 = x.shape[-1] // 2
    c = cos[positions].unsqueeze(2).to(x.dtype)
    s = sin[positions].unsqueeze(2).to(x.dtype)
    x1, x2 = x[..., :d2], x[..., d2:]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


class _AddColumnBlocks(torch.autograd.Function):
    """``y[..., off_i : off_i + n_i] += block_i`` in place for the blocks that are given (``y`` is a
    freshly computed projection nobody else holds).  The gradient of ``y`` is the incoming gradient
    itself and a block's gradient is its column slice of it: no copy of the wide tensor either way."""

    @staticmethod
    def forward(ctx, y, widths, *blocks):
        ctx.widths = widths
        off = 0
        for n, blk in zip(widths, blocks):
            if blk is not None:
                y[..., off:off + n].add_(blk)
            off += n
        ctx.mark_dirty(y)
        return y

    @staticmethod
    def backward(ctx, g):
        grads, off = [], 0
        for i, n in enumerate(ctx.widths):
            grads.append(g[..., off:off + n] if ctx.needs_input_grad[2 + i] else None)
            off += n
        return (g, None, *grads)


def add_column_blocks(y, blocks, widths):
    """Add ``blocks[i]`` (or nothing, for None) onto consecutive column ranges of ``y`` of the given
    widths (low-rank updates onto their parts of a grouped projection)."""
    return _AddColumnBlocks.apply(y, tuple(int(n) for n in widths), *blocks)


class _RopeQKV(torch.autograd.Function):
    """Packed ``[B,S,(Hq+2Hkv)*D]`` projection -> rotated q ``[B,S,Hq,D]``, rotated k and plain v
    ``[B,S,Hkv,D]`` in one launch; the backward rotates the three gradients back into one packed
    gradient in one launch (no strided copies, no concatenation)."""

    @staticmethod
    def forward(ctx, qkv, positions, cos, sin, hq, hkv, d):
        pos = positions.to(torch.long).contiguous()
        ctx.save_for_backward(pos, cos, sin)
        q, k, v = _raw().rope_qkv_split(qkv.contiguous(), pos, cos, sin, hq, hkv, d)
        return q, k, v

    @staticmethod
    def backward(ctx, gq, gk, gv):
        pos, cos, sin = ctx.saved_tensors
        g = _raw().rope_qkv_merge(gq.contiguous(), gk.contiguous(), gv.contiguous(), pos, cos, sin)
        return g, None, None, None, None, None, None


def rope_qkv_split(qkv, positions, cos, sin, hq: int, hkv: int, d: int):
    """``(rope(q), rope(k), v)`` from a packed q|k|v projection ``[B,S,(hq+2 hkv) d]``."""
    b, s = qkv.shape[:2]
    if (_fused(qkv) and qkv.dim() == 3 and d % 16 == 0 and cos.dtype == torch.float32 and cos.dim() == 2
            and positions.numel() == b * s):
        return _RopeQKV.apply(qkv, positions, cos.contiguous(), sin.contiguous(), hq, hkv, d)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    return (apply_rope(q.reshape(b, s, hq, d), positions, cos, sin),
            apply_rope(k.reshape(b, s, hkv, d), positions, cos, sin), v.reshape(b, s, hkv, d))


class _Attention(torch.autograd.Function):
    """Fused attention forward (+ log-sum-exp) and flash backward on the in-tree kernels."""

    @staticmethod
    def forward(ctx, q, k, v, causal, scale, q_offset):
        qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
        o, lse = _raw().attention_fwd_lse(qc, kc, vc, bool(causal), float(scale), int(q_offset))
        ctx.save_for_backward(qc, kc, vc, o, lse)
        ctx.args = (bool(causal), float(scale), int(q_offset))
        return o

    @staticmethod
    def backward(ctx, g):
        q, k, v, o, lse = ctx.saved_tensors
        causal, scale, q_offset = ctx.args
        dq, dk, dv = _raw().attention_bwd(q, k, v, o, g.contiguous(), lse, causal, scale, q_offset)
        return dq, dk, dv, None, None, None


def attention(q, k, v, causal: bool, scale: Optional[float] = None, q_offset: int = 0, key_padding_mask=None):
    """``[B,S,H,D]`` in and out; grouped KV heads are expanded; masks as in the oracle."""
    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
    if (key_padding_mask is None and _fused(q, k, v) and d in (64, 128) and h % hk == 0
            and (not causal or q_offset >= 0)):
        return _Attention.apply(q, k, v, causal, scale, q_offset)
    qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    if hk != h:
        kt = kt.repeat_interleave(h // hk, dim=1)
        vt = vt.repeat_interleave(h // hk, dim=1)
    plain_causal = causal and q_offset == sk - sq and sq == sk and key_padding_mask is None
    mask = None
    if not plain_causal and (causal or key_padding_mask is not None):
        mask = torch.ones(b, 1, sq, sk, dtype=torch.bool, device=q.device)
        if causal:
            qi = torch.arange(sq, device=q.device)[:, None] + q_offset
            mask = mask & (torch.arange(sk, device=q.device)[None, :] <= qi)
        if key_padding_mask is not None:
            mask = mask & key_padding_mask[:, None, None, :].bool()
    o = F.scaled_dot_product_attention(qt, kt, vt, attn_mask=mask, is_causal=plain_causal, scale=scale)
    if mask is not None:
        o = torch.nan_to_num(o)        # rows with every key masked
    return o.transpose(1, 2).contiguous()
