"""Differentiable GPU path: the ops a training step differentiates through, in the
tensors' own dtype on PyTorch-ROCm's library kernels.

The hand-written kernels in ``ops.hip`` are inference kernels (no backward) and
``ops.reference`` is the fp32 oracle (it upcasts every operand, which for a 7B decoder
means re-materialising each weight matrix in fp32 on every call).  A call on GPU tensors
that needs autograd comes here instead: bf16 GEMMs with fp32 accumulation through
``F.linear`` (hipBLASLt), ``F.scaled_dot_product_attention``, fp32 statistics only on the
activations of the norms.  Semantics match the oracle; rounding is that of bf16 training.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .reference import _ACTS


def rmsnorm(x, weight, eps: float = 1e-6):
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
    g, u = gate_up.chunk(2, dim=-1)
    return F.silu(g) * u


def apply_rope(x, positions, cos, sin):
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
