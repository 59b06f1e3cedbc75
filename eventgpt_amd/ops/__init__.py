"""Op front-end.

Every op has exactly two implementations:

* ``eventgpt_amd.ops.reference`` — plain PyTorch, fp32 internally.  Runs on CPU
  tensors and is the oracle the kernels are tested against.
* ``eventgpt_amd.ops.hip`` — the hand-written gfx950 kernels in the in-tree
  extension ``eventgpt_amd/_C.so``.  Runs on GPU tensors.

A GPU tensor never silently falls back to the reference path: if the extension
is missing, the call raises.  The one exception is a call that needs autograd
(training of adapters / LoRA), which uses differentiable PyTorch-ROCm ops —
the HIP kernels here are inference kernels.
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


def rmsnorm(x, weight, eps: float = 1e-6):
    if _use_hip(x, weight) and x.dtype == torch.bfloat16:
        return _hip().rmsnorm(x, weight, eps)
    return ref.rmsnorm(x, weight, eps)


def add_rmsnorm(x, residual, weight, eps: float = 1e-6):
    if _use_hip(x, residual, weight) and x.dtype == torch.bfloat16:
        return _hip().add_rmsnorm(x, residual, weight, eps)
    return ref.add_rmsnorm(x, residual, weight, eps)


def layernorm(x, weight, bias, eps: float = 1e-5):
    if _use_hip(x, weight, bias) and x.dtype == torch.bfloat16:
        return _hip().layernorm(x, weight, bias, eps)
    return ref.layernorm(x, weight, bias, eps)


def linear(x, weight, bias=None, act: Optional[str] = None, residual=None):
    if _use_hip(x, weight, bias, residual) and x.dtype == torch.bfloat16 \
            and weight.dtype == torch.bfloat16:
        return _hip().linear(x, weight, bias, act, residual)
    return ref.linear(x, weight, bias, act, residual)


def swiglu(gate_up):
    if _use_hip(gate_up) and gate_up.dtype == torch.bfloat16:
        return _hip().swiglu(gate_up)
    return ref.swiglu(gate_up)


def apply_rope(x, positions, cos, sin):
    if _use_hip(x) and x.dtype == torch.bfloat16:
        return _hip().apply_rope(x, positions, cos, sin)
    return ref.apply_rope(x, positions, cos, sin)


def attention(q, k, v, causal: bool, scale: Optional[float] = None, q_offset: int = 0,
              key_padding_mask=None):
    if _use_hip(q, k, v) and q.dtype == torch.bfloat16 and key_padding_mask is None \
            and _hip().attention_supported(q, k, v):
        return _hip().attention(q, k, v, causal, scale, q_offset)
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
