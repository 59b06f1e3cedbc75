"""Differentiable GPU path: the ops a training step differentiates through, in the
tensors' own dtype on PyTorch-ROCm's library kernels.

The hand-written kernels in ``ops.hip`` are inference kernels (no backward) and
``ops.reference`` is the fp32 oracle (it upcasts every operand, which for a 7B decoder
means re-materialising each weight matrix in fp32 on every call).  A call on GPU tensors
that needs autograd comes here instead: bf16 GEMMs with fp32 accumulation through
``F.linear`` (hipBLASLt), ``F.scaled_dot_product_attention``, fp32 statistics only on the
activations of the norms.  Semantics match the oracle; rounding is that of bf16 training.

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


def layernorm(x, weight, bias, eps: float = 1e-5):
    return F.layer_norm(x, (x.shape[-1],), weight.to(x.dtype), None if bias is None else bias.to(x.dtype), eps)


def linear(x, weight, bias=None, act: Optional[str] = None, residual=None):
    y = F.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))
    y = _ACTS[act](y)
    return y if residual is None else y + residual


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


def attention(q, k, v, causal: bool, scale: Optional[float] = None, q_offset: int = 0, key_padding_mask=None):
    """``[B,S,H,D]`` in and out; grouped KV heads are expanded; masks as in the oracle."""
    b, sq, h, d = q.shape
    sk, hk = k.shape[1], k.shape[2]
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
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
