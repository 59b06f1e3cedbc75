"""Plain-PyTorch implementations of every op in the framework.

These are the numerics oracle for the HIP kernels and the execution path on
CPU tensors.  They compute in fp32 internally and return the input dtype.
They are *not* an alternative GPU production path: on a GPU tensor
``eventgpt_amd.ops`` dispatches to the HIP extension and raises if it is absent.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------- norms
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * weight.float()).to(x.dtype)


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """``r = x + residual`` (rounded to the storage dtype), ``y = rmsnorm(r)``."""
    r = (x.float() + residual.float()).to(x.dtype)
    return rmsnorm(r, weight, eps), r


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
              eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x.float(), (x.shape[-1],), weight.float(),
                        None if bias is None else bias.float(), eps).to(x.dtype)


# ------------------------------------------------------------ activations
def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


_ACTS = {
    None: lambda t: t,
    "none": lambda t: t,
    "gelu": lambda t: F.gelu(t),
    "quick_gelu": quick_gelu,
    "silu": F.silu,
    "relu": F.relu,
}


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           act: Optional[str] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(x @ weightᵀ + bias) + residual`` with fp32 accumulation."""
    y = x.float() @ weight.float().t()
    if bias is not None:
        y = y + bias.float()
    y = _ACTS[act](y)
    if residual is not None:
        y = y + residual.float()
    return y.to(x.dtype)


def swiglu(gate_up: torch.Tensor) -> torch.Tensor:
    """``silu(gate) * up`` for ``gate_up = [..., 2*I]`` laid out as ``[gate | up]``."""
    g, u = gate_up.float().chunk(2, dim=-1)
    return (F.silu(g) * u).to(gate_up.dtype)


# -------------------------------------------------------------------- RoPE
def rope_tables(max_pos: int, dim: int, theta: float = 10000.0,
                device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 ``cos, sin`` tables of shape ``[max_pos, dim/2]``."""
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=device) / dim))
    ang = torch.arange(max_pos, dtype=torch.float32, device=device)[:, None] * inv[None, :]
    return ang.cos(), ang.sin()


def apply_rope(x: torch.Tensor, positions: torch.Tensor, cos: torch.Tensor,
               sin: torch.Tensor) -> torch.Tensor:
    """Rotate-half RoPE (LLaMA / HF convention).

    ``x``: ``[B, S, H, D]``; ``positions``: ``[B, S]`` (long).  Element ``i`` pairs
    with element ``i + D/2``.
    """
    d2 = x.shape[-1] // 2
    c = cos[positions].unsqueeze(2)  # [B,S,1,D/2]
    s = sin[positions].unsqueeze(2)
    xf = x.float()
    x1, x2 = xf[..., :d2], xf[..., d2:]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)


# --------------------------------------------------------------- attention
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool,
              scale: Optional[float] = None, q_offset: int = 0,
              key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Scaled dot-product attention on ``[B, S, H, D]`` tensors.

    ``q_offset`` is the absolute position of the first query row relative to
    key 0 (so a decode / verify step with ``Sq`` new rows against a cache of
    ``Sk`` keys uses ``q_offset = Sk - Sq``).  ``key_padding_mask``: ``[B, Sk]``
    bool, True = attend.
    """
    b, sq, h, d = q.shape
    sk = k.shape[1]
    scale = scale if scale is not None else 1.0 / math.sqrt(d)
    qf = q.float().permute(0, 2, 1, 3)
    kf = k.float().permute(0, 2, 1, 3)
    vf = v.float().permute(0, 2, 1, 3)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if causal:
        qi = torch.arange(sq, device=q.device)[:, None] + q_offset
        kj = torch.arange(sk, device=q.device)[None, :]
        s = s.masked_fill(kj > qi, float("-inf"))
    if key_padding_mask is not None:
        s = s.masked_fill(~key_padding_mask[:, None, None, :].bool(), float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p)  # rows with every key masked
    o = torch.matmul(p, vf)
    return o.permute(0, 2, 1, 3).contiguous().to(q.dtype)


# ----------------------------------------------------------- multimodal ops
def spatio_temporal_pool(features: torch.Tensor,
                         num_temporal_tokens: Optional[int] = None) -> torch.Tensor:
    """``[T, S, C]`` → ``[T' + S, C]``: per-frame means then per-token means.

    Behaviour of reference: model/EventChatModel.py:15-38.
    """
    if features.ndim != 3:
        raise ValueError("Input features should be a 3D tensor with shape (t, s, c)")
    t = features.shape[0]
    n_t = t if num_temporal_tokens is None else num_temporal_tokens
    ff = features.float()
    temporal = ff.mean(dim=1)
    if n_t > t:
        temporal = F.pad(temporal, (0, 0, 0, n_t - t))
    elif n_t < t:
        temporal = temporal[:n_t]
    spatial = ff.mean(dim=0)
    return torch.cat([temporal, spatial], dim=0).to(features.dtype)


def splice_embeddings(text_embeds: torch.Tensor, event_embeds: torch.Tensor,
                      event_pos: int) -> torch.Tensor:
    """Insert ``event_embeds [E, C]`` into ``text_embeds [L, C]`` before row ``event_pos``."""
    return torch.cat([text_embeds[:event_pos], event_embeds.to(text_embeds.dtype),
                      text_embeds[event_pos:]], dim=0)


# ----------------------------------------------------------------- sampling
def argmax_last(logits: torch.Tensor) -> torch.Tensor:
    return logits.float().argmax(dim=-1)


def sample_top_p(logits: torch.Tensor, temperature: float, top_p: float,
                 generator: Optional[torch.Generator] = None, top_k: int = 0) -> torch.Tensor:
    """Temperature, then top-k (0 = off; ties at the k-th value are kept), then nucleus
    sampling over the last dim; returns ``[...]`` long."""
    if temperature <= 0:
        return argmax_last(logits)
    scaled = logits.float() / temperature
    if 0 < top_k < scaled.shape[-1]:
        kth = scaled.topk(top_k, dim=-1).values[..., -1:]
        scaled = scaled.masked_fill(scaled < kth, float("-inf"))
    probs = torch.softmax(scaled, dim=-1)
    if top_p < 1.0:
        sp, si = probs.sort(dim=-1, descending=True)
        cum = sp.cumsum(dim=-1)
        drop = (cum - sp) > top_p  # keep the first token that crosses top_p
        sp = sp.masked_fill(drop, 0.0)
        sp = sp / sp.sum(dim=-1, keepdim=True)
        pick = torch.multinomial(sp.reshape(-1, sp.shape[-1]), 1, generator=generator)
        return si.reshape(-1, si.shape[-1]).gather(1, pick).reshape(logits.shape[:-1])
    pick = torch.multinomial(probs.reshape(-1, probs.shape[-1]), 1, generator=generator)
    return pick.reshape(logits.shape[:-1])


# ------------------------------------------------------- speculative decode
def verify_greedy(draft: torch.Tensor, target_argmax: torch.Tensor):
    """Greedy draft verification.

    ``draft``: ``[G]`` proposed tokens d1..dG.  ``target_argmax``: ``[G+1]``
    verifier argmax at the positions *preceding* each draft token plus the
    position after the last one.  Returns ``(n_accepted, next_token)`` where
    ``next_token`` is the verifier's own token at the first mismatch, or the
    bonus token when everything matched (reference:
    pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:569-637).
    """
    g = draft.shape[0]
    match = (draft == target_argmax[:g]).to(torch.int64)
    n_acc = int(torch.cumprod(match, dim=0).sum().item()) if g > 0 else 0
    return n_acc, int(target_argmax[n_acc].item())


# ------------------------------------------------------------------ metrics
def cosine_rows(a: torch.Tensor, b: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    af, bf = a.float(), b.float()
    return (af * bf).sum(-1) / (af.norm(dim=-1).clamp_min(eps) * bf.norm(dim=-1).clamp_min(eps))


def row_metrics(a: torch.Tensor, b: torch.Tensor, eps: float = 1e-8):
    """``(cosine, mean squared error)`` over the last dim, fp32."""
    af, bf = a.float(), b.float()
    return cosine_rows(af, bf, eps), (af - bf).pow(2).mean(-1)
