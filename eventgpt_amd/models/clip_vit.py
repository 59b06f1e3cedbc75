"""CLIP vision transformer (the frozen event tower).

Parameter names follow the HF ``CLIPVisionModel`` checkpoint layout
(``vision_model.embeddings.*``, ``vision_model.pre_layrnorm`` [sic],
``vision_model.encoder.layers.N.*``, ``vision_model.post_layernorm``).

The forward returns the *last encoder layer output for all tokens* (CLS
included, no post-layernorm) — that is what EventGPT consumes
(reference: model/EventChatModel.py:194-200).  All T frames of a sample are
run as one batch (M = T·577 rows per GEMM) instead of one forward per frame.

The patch embedding is a stride-14 14×14 convolution, i.e. a pure GEMM over
non-overlapping patches: ``[B·576, 588] × [588, 1024]``; no im2col buffer with
overlap is needed.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Tuple

import torch
from torch import nn

from .. import ops
from .config import ClipVisionConfig
from .llama import Linear


@dataclasses.dataclass
class VisionOutput:
    last_hidden_state: torch.Tensor
    pooler_output: Optional[torch.Tensor] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps)


class ClipEmbeddings(nn.Module):
    def __init__(self, cfg: ClipVisionConfig):
        super().__init__()
        self.cfg = cfg
        self.class_embedding = nn.Parameter(torch.randn(cfg.hidden_size) * 0.02)
        self.patch_embedding = nn.Conv2d(cfg.num_channels, cfg.hidden_size, cfg.patch_size,
                                         stride=cfg.patch_size, bias=False)
        self.position_embedding = nn.Embedding(cfg.num_tokens, cfg.hidden_size)

    def patches(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """``[B,C,H,W]`` → ``[B, N, C·P·P]`` in the conv weight's (c, ph, pw) order."""
        b, c, h, w = pixel_values.shape
        p = self.cfg.patch_size
        gh, gw = h // p, w // p
        x = pixel_values.reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(b, gh * gw, c * p * p)

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        w = self.patch_embedding.weight
        x = self.patches(pixel_values.to(w.dtype)).contiguous()
        pos = self.position_embedding.weight
        tok = ops.linear(x, w.reshape(w.shape[0], -1))            # [B, N, D]
        tok = tok + pos[1:][None].to(tok.dtype)
        cls = (self.class_embedding + pos[0]).to(tok.dtype)
        return torch.cat([cls.expand(tok.shape[0], 1, -1), tok], dim=1)


class ClipAttention(nn.Module):
    def __init__(self, cfg: ClipVisionConfig):
        super().__init__()
        d = cfg.hidden_size
        self.num_heads = cfg.num_attention_heads
        self.head_dim = d // self.num_heads
        self.q_proj = Linear(d, d)
        self.k_proj = Linear(d, d)
        self.v_proj = Linear(d, d)
        self.out_proj = Linear(d, d)

    def forward(self, x, residual):
        b, s, _ = x.shape
        shp = (b, s, self.num_heads, self.head_dim)
        q, k, v = self.q_proj(x).view(shp), self.k_proj(x).view(shp), self.v_proj(x).view(shp)
        o = ops.attention(q, k, v, causal=False)
        return self.out_proj(o.reshape(b, s, -1), residual=residual)


class ClipMLP(nn.Module):
    def __init__(self, cfg: ClipVisionConfig):
        super().__init__()
        self.act = cfg.hidden_act
        self.fc1 = Linear(cfg.hidden_size, cfg.intermediate_size)
        self.fc2 = Linear(cfg.intermediate_size, cfg.hidden_size)

    def forward(self, x, residual):
        return self.fc2(self.fc1(x, act=self.act), residual=residual)


class ClipEncoderLayer(nn.Module):
    def __init__(self, cfg: ClipVisionConfig):
        super().__init__()
        self.self_attn = ClipAttention(cfg)
        self.layer_norm1 = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = ClipMLP(cfg)
        self.layer_norm2 = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, x):
        x = self.self_attn(self.layer_norm1(x), residual=x)
        return self.mlp(self.layer_norm2(x), residual=x)


class ClipEncoder(nn.Module):
    def __init__(self, cfg: ClipVisionConfig):
        super().__init__()
        self.layers = nn.ModuleList(ClipEncoderLayer(cfg) for _ in range(cfg.num_hidden_layers))


class ClipVisionTransformer(nn.Module):
    def __init__(self, cfg: ClipVisionConfig):
        super().__init__()
        self.embeddings = ClipEmbeddings(cfg)
        self.pre_layrnorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.encoder = ClipEncoder(cfg)
        self.post_layernorm = LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, pixel_values, output_hidden_states: bool = False) -> VisionOutput:
        x = self.pre_layrnorm(self.embeddings(pixel_values))
        hs = [x] if output_hidden_states else None
        for layer in self.encoder.layers:
            x = layer(x)
            if output_hidden_states:
                hs.append(x)
        pooled = self.post_layernorm(x[:, 0])
        return VisionOutput(x, pooled, tuple(hs) if hs else None)


class ClipVisionModel(nn.Module):
    def __init__(self, cfg: Optional[ClipVisionConfig] = None):
        super().__init__()
        self.config = cfg or ClipVisionConfig.vit_l_14_336()
        self.vision_model = ClipVisionTransformer(self.config)

    def forward(self, pixel_values, output_hidden_states: bool = False) -> VisionOutput:
        return self.vision_model(pixel_values, output_hidden_states)
