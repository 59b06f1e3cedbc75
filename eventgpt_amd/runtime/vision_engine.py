"""VisionEngine — graph-captured event tower (ViT + projector + adaptor + pooling).

The module path (``models.clip_vit``) issues ~300 launches per tower pass from
Python, which is host-bound for a 5 ms workload.  This engine runs the same
weights as a fixed launch sequence on static buffers and captures it into one
HIP graph per frame count T:

  patchify (im2col, K padded to 640) -> patch GEMM -> fused CLS/pos/pre-LN ->
  24 x [LN, fused QKV GEMM(+bias), attention on strided q/k/v views,
        out_proj GEMM(+bias,+residual), LN, fc1 GEMM(+bias, quick-GELU),
        fc2 GEMM(+bias,+residual)] ->
  projector GEMM(+GELU) -> projector GEMM -> feature-adaptor GEMM ->
  spatio-temporal pooling  ->  [T + 577, hidden]

All T frames go through every GEMM as one M = T*577 batch.  The q/k/v weights
are packed into one [3*D, D] matrix (module parameters are re-pointed at views
of it).  Reference behaviour: per-frame ``visval_encode`` + ``feature_adaptor``
in a Python loop, then pooling (model/EventChatModel.py:333-343).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from ..ops.hip import ACT_CODES
from .engine import capture_tensor_mode, no_gc_during_capture, persistent_state


class VisionEngine:
    @persistent_state
    def __init__(self, model, use_graph: bool = True, tower=None, projector=None,
                 adaptor="auto", feature_layer: int = -1):
        """``tower``/``projector``/``adaptor`` default to the EventChat modules of
        ``model``; a video verifier passes its own.  ``feature_layer`` follows the
        HF ``hidden_states`` convention (-1 = after the last block, -2 = skip it)."""
        from ..ops import hip
        self.raw = hip.raw
        self.model = model
        if tower is None:
            tower = model.get_visual_tower().visual_tower
        self.cfg = tower.config
        vm = tower.vision_model
        self.device = model.device
        if self.device.type != "cuda" or model.dtype != torch.bfloat16:
            raise RuntimeError("VisionEngine needs a bf16 model on a GPU")
        self.use_graph = use_graph
        d = self.cfg.hidden_size
        if d % 64 or (d // self.cfg.num_attention_heads) not in (64, 128):
            raise ValueError("vision hidden size must be a multiple of 64 with head_dim 64/128")
        self.heads = self.cfg.num_attention_heads
        self.hd = d // self.heads
        self.act = ACT_CODES[self.cfg.hidden_act]
        emb = vm.embeddings
        w = emb.patch_embedding.weight.data
        k = w.shape[1] * w.shape[2] * w.shape[3]
        kp = (k + 63) // 64 * 64
        self.w_patch = torch.zeros(d, kp, dtype=torch.bfloat16, device=self.device)
        self.w_patch[:, :k] = w.reshape(d, k)
        self.cls = emb.class_embedding.data
        self.pos = emb.position_embedding.weight.data
        self.pre_ln = (vm.pre_layrnorm.weight.data, vm.pre_layrnorm.bias.data)
        self.eps = float(self.cfg.layer_norm_eps)
        self.layers = []
        with torch.no_grad():
            for lyr in vm.encoder.layers:
                at = lyr.self_attn
                w_qkv = torch.cat([at.q_proj.weight.data, at.k_proj.weight.data,
                                   at.v_proj.weight.data]).contiguous()
                b_qkv = torch.cat([at.q_proj.bias.data, at.k_proj.bias.data,
                                   at.v_proj.bias.data]).contiguous()
                for i, proj in enumerate((at.q_proj, at.k_proj, at.v_proj)):
                    proj.weight.data = w_qkv[i * d:(i + 1) * d]
                    proj.bias.data = b_qkv[i * d:(i + 1) * d]
                self.layers.append(dict(
                    ln1=(lyr.layer_norm1.weight.data, lyr.layer_norm1.bias.data),
                    w_qkv=w_qkv, b_qkv=b_qkv,
                    w_o=at.out_proj.weight.data, b_o=at.out_proj.bias.data,
                    ln2=(lyr.layer_norm2.weight.data, lyr.layer_norm2.bias.data),
                    w1=lyr.mlp.fc1.weight.data, b1=lyr.mlp.fc1.bias.data,
                    w2=lyr.mlp.fc2.weight.data, b2=lyr.mlp.fc2.bias.data))
        n_run = len(self.layers) + 1 + feature_layer if feature_layer < 0 else feature_layer
        if not 0 < n_run <= len(self.layers):
            raise ValueError(f"feature_layer {feature_layer} outside the tower's {len(self.layers)} blocks")
        self.layers = self.layers[:n_run]
        proj = model.get_model().visual_projector if projector is None else projector
        mods = proj if isinstance(proj, (list, tuple, torch.nn.Sequential)) else proj.children()
        self.proj = [(m.weight.data, m.bias.data) for m in mods if hasattr(m, "weight")]
        fa = getattr(model.get_model(), "feature_adaptor", None) if adaptor == "auto" else adaptor
        self.adaptor = (fa.weight.data, fa.bias.data) if fa is not None else None
        self._graphs: Dict[tuple, tuple] = {}

    # bias + residual + the next LayerNorm inside the split-K slab sum of out-proj / fc2 (see _forward)
    fuse_norms = True

    # ------------------------------------------------------------------ forward
    def _forward(self, px: torch.Tensor, pool: bool, n_temporal: Optional[int],
                 tower_only: bool = False) -> torch.Tensor:
        raw = self.raw
        T = px.shape[0]
        d, H, hd = self.cfg.hidden_size, self.heads, self.hd
        rows = raw.patchify(px, self.cfg.patch_size)
        tok = raw.linear(rows, self.w_patch, None, None, 0, False)
        n = self.cfg.num_tokens
        x = raw.vit_embed_ln(tok.view(T, n - 1, d), self.cls, self.pos, self.pre_ln[0],
                             self.pre_ln[1], self.eps)
        scale = hd ** -0.5
        inter = self.cfg.intermediate_size
        if self.fuse_norms and d <= 2048 and inter % 64 == 0:
            # The two row-parallel projections of a block stop at their fp32 split-K slabs, and the kernel that
            # sums them also adds bias and residual and applies the LayerNorm the next projection reads:
            # 5 launches per block (qkv, attention, out-proj, fc1, fc2 + two consumers = 7 with them) instead of
            # 7 plus a split-K reduction, and the narrow out-projection may split its contraction too.
            x = x.view(T * n, d)
            y = raw.layernorm(x, self.layers[0]["ln1"][0], self.layers[0]["ln1"][1], self.eps)
            for i, l in enumerate(self.layers):
                qkv = raw.linear(y, l["w_qkv"], l["b_qkv"], None, 0, False).view(T, n, 3, H, hd)
                o = raw.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], False, scale, 0)
                x, y = raw.reduce_residual_layernorm(raw.linear_partial(o.view(T * n, d), l["w_o"]), l["b_o"], x,
                                                     l["ln2"][0], l["ln2"][1], self.eps)
                y = raw.linear(y, l["w1"], l["b1"], None, self.act, False)
                nxt = self.layers[i + 1]["ln1"] if i + 1 < len(self.layers) else (None, None)
                x, y = raw.reduce_residual_layernorm(raw.linear_partial(y, l["w2"]), l["b2"], x, nxt[0], nxt[1],
                                                     self.eps)
            f = x.view(T, n, d)
        else:
            for l in self.layers:
                y = raw.layernorm(x, l["ln1"][0], l["ln1"][1], self.eps)
                qkv = raw.linear(y, l["w_qkv"], l["b_qkv"], None, 0, False).view(T, n, 3, H, hd)
                o = raw.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], False, scale, 0)
                x = raw.linear(o.view(T, n, d), l["w_o"], l["b_o"], x, 0, False)
                y = raw.layernorm(x, l["ln2"][0], l["ln2"][1], self.eps)
                y = raw.linear(y, l["w1"], l["b1"], None, self.act, False)
                x = raw.linear(y, l["w2"], l["b2"], x, 0, False)
            f = x
        if tower_only:
            return f
        for i, (w, b) in enumerate(self.proj):
            f = raw.linear(f, w, b, None, ACT_CODES["gelu"] if i + 1 < len(self.proj) else 0, False)
        if self.adaptor is not None:
            f = raw.linear(f, self.adaptor[0], self.adaptor[1], None, 0, False)
        if pool:
            return raw.st_pool(f, T if n_temporal is None else n_temporal)
        return f

    @torch.no_grad()
    def encode(self, frames, pool: bool = True, n_temporal: Optional[int] = None,
               tower_only: bool = False) -> torch.Tensor:
        """``frames``: ``[T,3,S,S]`` tensor or list of ``[3,S,S]``.  Returns pooled
        ``[T'+N, hidden]`` (or per-frame ``[T, N, hidden]`` with ``pool=False``).
        ``tower_only``: stop after the (frozen) ViT blocks and return ``[T, N, vision_hidden]`` --
        what a training step feeds its trainable projector."""
        if isinstance(frames, (list, tuple)):
            frames = torch.stack([f.to(self.device) for f in frames])
        px = frames.to(self.device, torch.bfloat16).contiguous()
        if not self.use_graph:
            return self._forward(px, pool, n_temporal, tower_only)
        key = (tuple(px.shape), pool, n_temporal, tower_only)
        entry = self._graphs.get(key)
        if entry is None:
            with capture_tensor_mode():
                static_in = px.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._forward(static_in, pool, n_temporal, tower_only)     # warm-up
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with no_gc_during_capture(), torch.cuda.graph(g):
                    static_out = self._forward(static_in, pool, n_temporal, tower_only)
            entry = (g, static_in, static_out)
            self._graphs[key] = entry
        g, static_in, static_out = entry
        static_in.copy_(px)
        g.replay()
        return static_out.clone()

    def encode_tower(self, frames) -> torch.Tensor:
        """The frozen tower alone: ``[T, N, vision_hidden]`` (one graph replay)."""
        return self.encode(frames, pool=False, tower_only=True)
