"""Inference forward of the hidden-state adapters on the in-tree gfx950 kernels.

Inside a speculative-decoding round the adapter is the draft head: ``[1..22, 4096]`` drafter
states in, aligned states out, every round.  As ``nn.Module``s that is a chain of small
library launches per round (LayerNorm, GEMM, SDPA with a 512-wide head that no fused kernel
takes).  ``AdapterKernels`` runs the same computation through ``evgpt::`` ops only:

* LayerNorm (+ fused GELU where the adapter has ``LN -> GELU``) -- ``layernorm`` kernel;
* every Linear -- the MFMA GEMM / weight-streaming GEMV with bias, GELU and residual in the
  epilogue; the residual scale ``alpha`` is folded into the up-projection weights once;
* multi-head attention with head_dim 512 (8 heads x 512 = 4096) -- ``attn_d512`` (one wave
  per query row); heads of 64 / 128 take the fused prefill attention kernel;
* the few remaining element-wise steps (position embedding add, sigmoid gate, final blend)
  are plain tensor expressions.

Nothing allocates through a library workspace or synchronises, so the whole draft head is
recorded into the hipGraph of a speculative-decoding iteration (``runtime/spec_decode.py``).
Weights are converted to bf16 once; the module itself (fp32, autograd) stays the training path.

Reference: pipeline/adapter_train/hidden_adapter.py:40-1199 (the modules), used from the SD
loop at pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:996-1001.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import adapters as A

GELU, RELU = 1, 4


def _bf(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().to(torch.bfloat16).contiguous()


class _Lin:
    def __init__(self, lin: nn.Linear, scale: float = 1.0):
        self.w = _bf(lin.weight * scale)
        self.b = _bf(lin.bias * scale) if lin.bias is not None else None


class _LN:
    def __init__(self, ln):
        self.identity = not isinstance(ln, nn.LayerNorm)
        if not self.identity:
            self.w, self.b, self.eps = _bf(ln.weight), _bf(ln.bias), float(ln.eps)


class AdapterKernels:
    """``kernels(h [S,H] bf16[, verifier_h [S,H]]) -> aligned [S,H] bf16`` for one sequence."""

    SUPPORTED = (A.HiddenStateAdapter, A.MultiLayerBottleneckAdapter, A.WideBottleneckAdapter,
                 A.AttentionAdapter, A.EAGLEStyleAdapter, A.FusedEAGLEAdapter)

    @classmethod
    def supports(cls, adapter) -> bool:
        if not isinstance(adapter, cls.SUPPORTED):
            return False
        if isinstance(adapter, A.EAGLEStyleAdapter) and adapter.use_token_embed:
            return False
        h = getattr(adapter, "hidden_dim", None) or adapter.config.hidden_dim
        return h % 64 == 0

    def __init__(self, adapter: nn.Module):
        from ..ops import hip
        self.raw = hip.raw
        if not self.supports(adapter):
            raise TypeError(f"{type(adapter).__name__} has no kernel forward")
        self.kind = type(adapter)
        a = adapter
        if isinstance(a, (A.HiddenStateAdapter, A.MultiLayerBottleneckAdapter)):
            blocks = a.layers if isinstance(a, A.HiddenStateAdapter) else a.blocks
            self.bottlenecks = [self._prep_bottleneck(b) for b in blocks]
            self.final = _LN(a.final_norm)
        elif isinstance(a, A.WideBottleneckAdapter):
            self.input_norm = _LN(a.input_norm)
            self.wide = [(_Lin(blk[0]), _LN(blk[1]), _Lin(blk[4], float(al))) for blk, al in zip(a.blocks, a.alphas)]
            self.final = _LN(a.final_norm)
        else:
            self.heads = a.num_heads
            self.blocks = [self._prep_block(l) for l in a.layers]
            self.alpha = float(a.alpha)
            if isinstance(a, A.AttentionAdapter):
                self.out_norm, self.out_proj = _LN(a.output_proj[0]), _Lin(a.output_proj[1], self.alpha)
                self.causal = False
            else:
                self.out_norm, self.out_proj = _LN(a.output_norm), _Lin(a.output_proj, self.alpha)
                self.pos = _bf(a.pos_embed[0])
                self.causal = True
            if isinstance(a, A.EAGLEStyleAdapter):
                self.input_norm = _LN(a.input_norm)
            if isinstance(a, A.FusedEAGLEAdapter):
                self.fusion_proj, self.fusion_norm, self.gate = _Lin(a.fusion_proj), _LN(a.fusion_norm), _Lin(a.gate[0])

    # ------------------------------------------------------------------ pieces
    def _prep_bottleneck(self, b: A.BottleneckAdapter):
        act = GELU if isinstance(b.act, nn.GELU) else RELU
        return (_LN(b.layer_norm), _Lin(b.down_proj), act, _Lin(b.up_proj, float(b.alpha_param)))

    def _prep_block(self, layer):
        at = layer["attn"]
        return {"attn_norm": _LN(layer["attn_norm"]), "in_w": _bf(at.in_proj_weight), "in_b": _bf(at.in_proj_bias),
                "out": _Lin(at.out_proj), "ffn_norm": _LN(layer["ffn_norm"]),
                "ffn1": _Lin(layer["ffn"][0]), "ffn2": _Lin(layer["ffn"][3])}

    def _ln(self, ln: _LN, x, act: int = 0):
        if ln.identity:
            return x
        return self.raw.layernorm(x, ln.w, ln.b, ln.eps, act)

    def _lin(self, lin: _Lin, x, act: int = 0, res=None):
        from ..ops import hip
        return hip.linear(x, lin.w, lin.b, {0: None, GELU: "gelu", RELU: "relu"}[act], res)

    def _attention(self, qkv: torch.Tensor, causal: bool) -> torch.Tensor:
        s, e3 = qkv.shape
        e = e3 // 3
        hd = e // self.heads
        v5 = qkv.view(1, s, 3, self.heads, hd)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        scale = 1.0 / math.sqrt(hd)
        if hd == 512:
            return self.raw.attn_d512(q, k, v, causal, scale).view(s, e)
        if hd in (64, 128):
            return self.raw.attention(q, k, v, causal, scale, 0).view(s, e)
        o = torch.nn.functional.scaled_dot_product_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal)
        return o.transpose(1, 2).reshape(s, e)

    def _run_blocks(self, x: torch.Tensor) -> torch.Tensor:
        from ..ops import hip
        for b in self.blocks:
            qkv = hip.linear(self._ln(b["attn_norm"], x), b["in_w"], b["in_b"])
            x = self._lin(b["out"], self._attention(qkv, self.causal), res=x)
            f = self._lin(b["ffn1"], self._ln(b["ffn_norm"], x), act=GELU)
            x = self._lin(b["ffn2"], f, res=x)
        return x

    # ----------------------------------------------------------------- forward
    @torch.no_grad()
    def __call__(self, h: torch.Tensor, verifier_h: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = h.to(torch.bfloat16).reshape(-1, h.shape[-1]).contiguous()
        k = self.kind
        if k in (A.HiddenStateAdapter, A.MultiLayerBottleneckAdapter):
            x = h
            for ln, down, act, up in self.bottlenecks:
                x = self._lin(up, self._lin(down, self._ln(ln, x), act=act), res=x)
            return self._ln(self.final, x)
        if k is A.WideBottleneckAdapter:
            x = self._ln(self.input_norm, h)
            for w1, ln, w2 in self.wide:
                x = self._lin(w2, self._ln(ln, self._lin(w1, x), act=GELU), res=x)
            return self._ln(self.final, x)
        s = h.shape[0]
        if k is A.AttentionAdapter:
            base, x = h, h
        elif k is A.EAGLEStyleAdapter:
            base = h
            x = self._ln(self.input_norm, h) + self.pos[:s]
        else:
            if verifier_h is None:
                raise ValueError("L5F needs the verifier hidden states as its second input")
            v = verifier_h.to(torch.bfloat16).reshape(-1, h.shape[-1]).contiguous()
            both = torch.cat([h, v], dim=-1)
            g = torch.sigmoid(self._lin(self.gate, both).float())
            fused = (g * h.float() + (1 - g) * v.float()).to(torch.bfloat16)
            x = self._ln(self.fusion_norm, self._lin(self.fusion_proj, both, res=fused)) + self.pos[:s]
            base = v
        x = self._run_blocks(x)
        # base + alpha * (W y + b - base): alpha is folded into W and b
        keep = (base.float() * (1.0 - self.alpha)).to(torch.bfloat16)
        return self._lin(self.out_proj, self._ln(self.out_norm, x), res=keep)
