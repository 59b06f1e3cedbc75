"""Python face of the in-tree gfx950 extension (``eventgpt_amd/_C.so``).

Importing this module loads the shared library; if it is missing the import
raises, so a GPU op can never quietly run on a fallback.  Build it with
``python -m eventgpt_amd.build`` (hipcc, ``--offload-arch=gfx950``).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

# EVGPT_SO picks another build of the same sources (a tuning variant linked by
# ``python -m eventgpt_amd.build --extra ... --out ...``)
_SO = os.environ.get("EVGPT_SO") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_C.so")
if not os.path.exists(_SO):
    raise RuntimeError(
        f"{_SO} not found: the HIP extension is not built. Run `python -m eventgpt_amd.build` "
        "(needs hipcc; cross-compiles for gfx950 without a GPU).")
torch.ops.load_library(_SO)
_ops = torch.ops.evgpt

ACT_CODES = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "silu": 3, "relu": 4}


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def rmsnorm(x, weight, eps=1e-6):
    return _ops.rmsnorm(_c(x), _c(weight), eps)


def add_rmsnorm(x, residual, weight, eps=1e-6):
    return _ops.add_rmsnorm(_c(x), _c(residual), _c(weight), eps)


def layernorm(x, weight, bias, eps=1e-5):
    return _ops.layernorm(_c(x), _c(weight), None if bias is None else _c(bias), eps)


def _pad_k(x: torch.Tensor, w: torch.Tensor, mult: int = 64):
    """Zero-pad the contraction dim of both operands to a multiple of ``mult``."""
    k = x.shape[-1]
    pad = (-k) % mult
    if pad == 0:
        return x, w
    return torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(w, (0, pad))


def linear(x, weight, bias=None, act=None, residual=None, out_f32: bool = False):
    x, weight = _c(x), _c(weight)
    if x.shape[-1] % 8:       # the kernels move 16-byte chunks; any multiple of 8 runs as it is
        x, weight = _pad_k(x, weight)
    return _ops.linear(x, weight, None if bias is None else _c(bias),
                       None if residual is None else _c(residual), ACT_CODES[act], out_f32)


def swiglu(gate_up):
    return _ops.swiglu(_c(gate_up))


def apply_rope(x, positions, cos, sin):
    return _ops.rope(_c(x), _c(positions.to(torch.long)), cos, sin)


def attention_supported(q, k, v) -> bool:
    d = q.shape[-1]
    return d in (64, 128) and q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1 \
        and all(s % 8 == 0 for t in (q, k, v) for s in t.stride()[1:3])


def attention(q, k, v, causal, scale=None, q_offset=0):
    scale = scale if scale is not None else 1.0 / math.sqrt(q.shape[-1])
    return _ops.attention(q, k, v, causal, scale, q_offset)


def spatio_temporal_pool(features, num_temporal_tokens: Optional[int] = None):
    n_t = features.shape[0] if num_temporal_tokens is None else num_temporal_tokens
    return _ops.st_pool(_c(features), n_t)


def argmax_last(logits):
    return _ops.argmax_rows(_c(logits))


# engine-level in-place ops are used through ``raw``
raw = _ops
