"""Op front-end.

* ``eventgpt_amd.ops.hip`` — the hand-written gfx950 kernels in the in-tree
  extension ``eventgpt_amd/_C.so``: every inference call on GPU tensors.  If the
  extension is missing the call raises; there is no quiet fallback.
* ``eventgpt_amd.ops.reference`` — plain PyTorch, fp32 internally: CPU tensors, and
  the oracle the kernels are tested against.
* ``eventgpt_amd.ops.differentiable`` — a GPU call that needs autograd (training of the
  projector, adapters, LoRA, the decoder): PyTorch-ROCm library ops in the tensors' own
  dtype, since the HIP kernels have no backward and the oracle upcasts every operand.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import reference as ref


def _use_hip(*tensors) -> bool:
    t0 = tensors[0]
    if not t0.is_cuda:
        return False
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False
    return True


def _hip():
    from . import hip
    return hip


def _train(*tensors) -> bool:
    """GPU tensors in a call that autograd records: take the differentiable library path."""
    t0 = tensors[0]
    return (t0.is_cuda and torch.is_grad_enabled()
            and any(t is not None and t.requires_grad for t in tensors))


def _diff():
    from . import differentiable
    return differentiable


def rmsnorm(x, weight, eps: float = 1e-6):
    if _use_hip(x, weight) and x.dtype == torch.bfloat16:
        return _hip().rmsnorm(x, weight, eps)
    if _train(x, weight):
        return _diff().rmsnorm(x, weight, eps)
    return ref.rmsnorm(x, weight, eps)


def add_rmsnorm(x, residual, weight, eps: float = 1e-6):
    if _use_hip(x, residual, weight) and x.dtype == torch.bfloat16:
        return _hip().add_rmsnorm(x, residual, weight, eps)
    if _train(x, residual, weight):
        return _diff().add_rmsnorm(x, residual, weight, eps)
    return ref.add_rmsnorm(x, residual, weight, eps)


def layernorm(x, weight, bias, eps: float = 1e-5):
    if _use_hip(x, weight, bias) and x.dtype == torch.bfloat16:
        return _hip().layernorm(x, weight, bias, eps)
    if _train(x, weight, bias):
        return _diff().layernorm(x, weight, bias, eps)
    return ref.layernorm(x, weight, bias, eps)


def linear(x, weight, bias=None, act: Optional[str] = None, residual=None):
    if _use_hip(x, weight, bias, residual) and x.dtype == torch.bfloat16 \
            and weight.dtype == torch.bfloat16:
        return _hip().linear(x, weight, bias, act, residual)
    if _train(x, weight, bias, residual):
        return _diff().linear(x, weight, bias, act, residual)
    return ref.linear(x, weight, bias, act, residual)


def swiglu(gate_up):
    if _use_hip(gate_up) and gate_up.dtype == torch.bfloat16:
        return _hip().swiglu(gate_up)
    if _train(gate_up):
        return _diff().swiglu(gate_up)
    return ref.swiglu(gate_up)


def apply_rope(x, positions, cos, sin):
    if _use_hip(x) and x.dtype == torch.bfloat16:
        return _hip().apply_rope(x, positions, cos, sin)
    if _train(x):
        return _diff().apply_rope(x, positions, cos, sin)
    return ref.apply_rope(x, positions, cos, sin)


def lowrank_delta(x, a, b, scale: float):
    """``(scale * x A^T) B^T`` with A ``[r, K]`` and B ``[N, r]``: a low-rank adapter's update."""
    if _use_hip(x, a, b) and x.dtype == a.dtype == b.dtype == torch.bfloat16 and a.shape[0] <= 16 \
            and a.shape[1] % 8 == 0 and a.is_contiguous() and x.numel() > 0:
        return _hip().linear(_hip().raw.lowrank_out(x.contiguous(), a, float(scale)), b)
    if _train(x, a, b):
        return _diff().lowrank_delta(x, a, b, scale)
    return linear(linear(x, a) * scale, b)


def lora_projection(x, weight, adapters, scale: float, drop_p: float = 0.0, residual=None):
    """``x W^T (+ residual)`` of a frozen projection with low-rank adapters ``[(column offset, A, B), ...]`` on
    column ranges of its output and input dropout ``drop_p`` per adapter, as ONE autograd node on the
    in-tree kernels (``ops.differentiable._LoRAProjection``; the same launches unrecorded when autograd is off)
    -- or None when those kernels do not take these operands (CPU tensors, other dtypes, a trainable base):
    the caller then composes the same thing from ``linear`` / ``lowrank_delta`` / ``add_column_blocks``."""
    if adapters and x.is_cuda and _diff().lora_projection_ok(x, weight, adapters, residual):
        return _diff().lora_projection(x, weight, adapters, scale, drop_p, residual)
    return None


def add_column_blocks(y, blocks, widths):
    """``y[..., off_i : off_i + widths[i]] += blocks[i]`` for the blocks that are not None; ``y`` is a
    freshly computed tensor (updated in place).  Returns ``y``."""
    if torch.is_grad_enabled() and (y.requires_grad or any(b is not None and b.requires_grad for b in blocks)):
        return _diff().add_column_blocks(y, blocks, widths)
    off = 0
    for n, blk in zip(widths, blocks):
        if blk is not None:
            y[..., off:off + n].add_(blk)
        off += n
    return y


def rope_qkv_split(qkv, positions, cos, sin, hq: int, hkv: int, d: int):
    """``(rope(q) [B,S,hq,d], rope(k), v [B,S,hkv,d])`` from a packed q|k|v projection
    ``[B,S,(hq+2 hkv) d]`` (the output of one GEMM over grouped q/k/v weights)."""
    b, s = qkv.shape[:2]
    fits = qkv.dim() == 3 and d % 16 == 0 and cos.dtype == torch.float32 and cos.dim() == 2 \
        and positions.numel() == b * s and qkv.dtype == torch.bfloat16
    if _use_hip(qkv) and fits:
        q, k, v = _hip().raw.rope_qkv_split(qkv.contiguous(), positions.to(torch.long).contiguous(),
                                            cos.contiguous(), sin.contiguous(), hq, hkv, d)
        return q, k, v
    if _train(qkv):
        return _diff().rope_qkv_split(qkv, positions, cos, sin, hq, hkv, d)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    return (apply_rope(q.reshape(b, s, hq, d), positions, cos, sin),
            apply_rope(k.reshape(b, s, hkv, d), positions, cos, sin), v.reshape(b, s, hkv, d))


def attention(q, k, v, causal: bool, scale: Optional[float] = None, q_offset: int = 0,
              key_padding_mask=None):
    if _use_hip(q, k, v) and q.dtype == torch.bfloat16 and key_padding_mask is None \
            and _hip().attention_supported(q, k, v):
        return _hip().attention(q, k, v, causal, scale, q_offset)
    if _train(q, k, v):
        return _diff().attention(q, k, v, causal, scale, q_offset, key_padding_mask)
    return ref.attention(q, k, v, causal, scale, q_offset, key_padding_mask)


def spatio_temporal_pool(features, num_temporal_tokens: Optional[int] = None):
    if _use_hip(features) and features.dtype == torch.bfloat16:
        return _hip().spatio_temporal_pool(features, num_temporal_tokens)
    return ref.spatio_temporal_pool(features, num_temporal_tokens)


def argmax_last(logits):
    if _use_hip(logits) and logits.dtype in (torch.bfloat16, torch.float32):
        return _hip().argmax_last(logits)
    return ref.argmax_last(logits)


rope_tables = ref.rope_tables
quick_gelu = ref.quick_gelu
sample_top_p = ref.sample_top_p
splice_embeddings = ref.splice_embeddings
verify_greedy = ref.verify_greedy
def row_metrics(a, b, eps: float = 1e-8):
    """``(cosine, mse)`` per row of two equally shaped tensors: one fused pass on a GPU."""
    if _use_hip(a) and b.is_cuda and a.dtype == b.dtype and a.dtype in (torch.float32, torch.bfloat16) \
            and a.shape == b.shape and a.shape[-1] % 8 == 0 and not (a.requires_grad or b.requires_grad):
        return _hip().raw.row_metrics(a.contiguous(), b.contiguous(), float(eps))
    return ref.row_metrics(a, b, eps)


def cosine_rows(a, b, eps: float = 1e-8):
    return row_metrics(a, b, eps)[0]
