"""Run the ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.GELU`` / ``nn.Dropout`` layers of an existing module
tree on the in-tree kernels.

The hidden-state adapters (``align/adapters.py``) keep the reference's module structure --
``nn.Linear`` / ``nn.Sequential`` stacks with its parameter names, so its checkpoints load
(pipeline/adapter_train/hidden_adapter.py:40-560) -- and train as fp32 parameters under bf16
autocast.  Every ``F.linear`` of such a tree goes to the vendor GEMM library.  Inside
``InTreeLinear()`` the same calls run ``ops.differentiable.linear`` instead: operands are cast to
bf16 exactly as autocast would (the casts are differentiable, so the weight gradient arrives at the
fp32 parameter), forward / dgrad / wgrad are the MFMA GEMM of ``csrc/kernels/gemm.hip``.  Calls the
in-tree kernel does not take (CPU tensors, dimensions that are no multiple of 8, a handful of rows)
fall through to the library untouched.

The same mode routes ``F.layer_norm``, ``F.gelu`` and ``F.dropout`` of CUDA activations to
``ops.differentiable.layernorm / gelu / dropout`` (bf16 activations, fp32 statistics; LayerNorm
parameters keep their fp32 gradients), so a training step of an attention adapter launches no
library normalisation / activation / dropout kernel.  The adapters' head_dim-512 attention calls
``ops.differentiable.attention_d512`` itself (``align/adapters.py: SelfAttention``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

from . import differentiable as diff


def _eligible(x, w, b) -> bool:
    if not (isinstance(x, torch.Tensor) and isinstance(w, torch.Tensor)):
        return False
    if not (x.is_cuda and w.is_cuda and w.dim() == 2 and x.dim() >= 2):
        return False
    if x.dtype not in (torch.bfloat16, torch.float32) or w.dtype not in (torch.bfloat16, torch.float32):
        return False
    n, k = w.shape
    if n % 8 or k % 8 or n < 8 or k < 8 or x.shape[-1] != k:
        return False
    if b is not None and not (isinstance(b, torch.Tensor) and b.is_cuda and b.numel() == n):
        return False
    return x.numel() // k >= 16 and diff._raw() is not None


class InTreeLinear(TorchFunctionMode):
    """``with InTreeLinear(): loss = model(...)`` -- see the module docstring."""

    def __init__(self):
        super().__init__()
        self.routed = 0          # calls that ran on the in-tree kernel (tests and benches read this)
        self.passed = 0          # calls left to the library

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.linear:
            x, w = args[0], args[1]
            b = args[2] if len(args) > 2 else kwargs.get("bias")
            if _eligible(x, w, b):
                self.routed += 1
                bf = torch.bfloat16
                return diff.linear(x.to(bf), w.to(bf), None if b is None else b.to(bf))
            self.passed += 1
        elif func is F.layer_norm:
            x = args[0]
            shape = args[1] if len(args) > 1 else kwargs.get("normalized_shape")
            w = args[2] if len(args) > 2 else kwargs.get("weight")
            b = args[3] if len(args) > 3 else kwargs.get("bias")
            eps = args[4] if len(args) > 4 else kwargs.get("eps", 1e-5)
            if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32)
                    and w is not None and len(tuple(shape)) == 1 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192
                    and x.numel() // x.shape[-1] >= 16 and diff._raw() is not None):
                self.routed += 1
                return diff.layernorm(x.to(torch.bfloat16), w, b, eps)
            self.passed += 1
        elif func is F.gelu:
            x = args[0]
            approx = args[1] if len(args) > 1 else kwargs.get("approximate", "none")
            if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.bfloat16 and approx == "none"
                    and x.numel() % 8 == 0 and x.numel() > 0 and diff._raw() is not None):
                self.routed += 1
                return diff.gelu(x)
            self.passed += 1
        elif func is F.dropout:
            x = args[0]
            p = args[1] if len(args) > 1 else kwargs.get("p", 0.5)
            training = args[2] if len(args) > 2 else kwargs.get("training", True)
            inplace = args[3] if len(args) > 3 else kwargs.get("inplace", False)
            if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.bfloat16 and training and 0 < p < 1
                    and not inplace and x.numel() % 8 == 0 and x.numel() > 0 and diff._raw() is not None):
                self.routed += 1
                return diff.dropout(x, p, True)
            self.passed += 1
        return func(*args, **kwargs)
