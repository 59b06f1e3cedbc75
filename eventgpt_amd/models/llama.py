"""LLaMA / Vicuna decoder.

Module and parameter names follow the HF checkpoint layout
(``model.embed_tokens``, ``model.layers.N.self_attn.{q,k,v,o}_proj``,
``model.layers.N.mlp.{gate,up,down}_proj``, ``…layernorm``, ``model.norm``,
``lm_head``) so published state dicts load without a key map.

This is the *module* path: general (arbitrary batch, padding masks, dynamic
``(k, v)`` tuple caches, hidden-state outputs, autograd) and the surface the
harness code pokes (``model.model(inputs_embeds=…, past_key_values=…)``,
reference: pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:389-397).  Every
tensor op goes through ``eventgpt_amd.ops`` and therefore runs the HIP kernels
on GPU tensors.  The latency-critical serving path is
``eventgpt_amd.runtime.engine.GenerationEngine``, which reuses these weights in
packed form with a static KV cache and a captured decode graph.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import ops


@dataclasses.dataclass
class DecoderOutput:
    last_hidden_state: torch.Tensor
    past_key_values: Optional[Tuple[Tuple[torch.Tensor, torch.Tensor], ...]] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None

    def __getitem__(self, i):
        return (self.last_hidden_state, self.past_key_values, self.hidden_states)[i]


@dataclasses.dataclass
class CausalLMOutput:
    logits: torch.Tensor
    past_key_values: Optional[tuple] = None
    hidden_states: Optional[tuple] = None
    loss: Optional[torch.Tensor] = None


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.rmsnorm(x, self.weight, self.variance_epsilon)


class Linear(nn.Linear):
    """``nn.Linear`` whose forward goes through the framework GEMM/GEMV."""

    def forward(self, x, act: Optional[str] = None, residual=None):
        return ops.linear(x, self.weight, self.bias, act, residual)


def group_frozen_weights(mods) -> Optional[torch.Tensor]:
    """Re-home the weights of frozen projections that read the same input in ONE ``[sum N, K]``
    matrix: each module's ``weight`` becomes a row-slice view of it (same values, same shapes, no
    extra memory), and a forward over the group is one GEMM of the combined width instead of one
    per projection -- a few hundred rows against 4096 columns fill barely one round of tiles, the
    same rows against 12288 or 22016 columns run 1.3-1.5x faster per FLOP -- and the input gradient
    one GEMM over the combined contraction instead of one per projection plus their sum.  Only for
    plain bias-free bf16 GPU layers whose weights do not train.  Returns the matrix, or None."""
    ws = [getattr(m, "weight", None) for m in mods]
    for m, w in zip(mods, ws):
        if type(m) is not Linear or m.bias is not None or w is None or w.requires_grad or not w.is_cuda \
                or w.dtype != torch.bfloat16 or w.dim() != 2 or w.shape[1] != ws[0].shape[1] or w.shape[0] % 64:
            return None
    cat = torch.cat([w.detach() for w in ws], 0)
    r = 0
    for w in ws:
        w.data = cat[r:r + w.shape[0]]
        w._in_group = True
        r += w.shape[0]
    cat._members = ws
    return cat


def group_intact(cat: Optional[torch.Tensor], mods, x: torch.Tensor) -> bool:
    """The group matrix still IS these modules' weights (none re-assigned, moved, wrapped or
    unfrozen since) and fits the input."""
    if cat is None or not x.is_cuda or x.dtype != cat.dtype:
        return False
    at, row_bytes = cat.data_ptr(), cat.shape[1] * cat.element_size()
    for m in mods:
        w = getattr(m, "weight", None)
        if type(m) is not Linear or w is None or w.requires_grad or w.data_ptr() != at or w.dtype != cat.dtype:
            return False
        at += w.shape[0] * row_bytes
    return True


class LlamaAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        h = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.num_kv_heads = config.num_key_value_heads
        self.head_dim = h // self.num_heads
        self.q_proj = Linear(h, self.num_heads * self.head_dim, bias=False)
        self.k_proj = Linear(h, self.num_kv_heads * self.head_dim, bias=False)
        self.v_proj = Linear(h, self.num_kv_heads * self.head_dim, bias=False)
        self.o_proj = Linear(self.num_heads * self.head_dim, h, bias=False)

    def _qkv_bases(self):
        """q / k / v as plain layers: a low-rank wrapper (``.base`` + ``lora_delta``) counts as its base."""
        return tuple(getattr(m, "base", m) for m in (self.q_proj, self.k_proj, self.v_proj))

    def group_frozen(self) -> bool:
        """``group_frozen_weights`` over q / k / v: one projection GEMM, split and rotated by one launch."""
        self._w_qkv = group_frozen_weights(self._qkv_bases())
        return self._w_qkv is not None

    def _qkv_with_adapters(self, x, w, bases):
        """``x W_qkv^T`` plus the low-rank updates of wrapped q / k / v on their column ranges."""
        mods = (self.q_proj, self.k_proj, self.v_proj)
        widths = [m.weight.shape[0] for m in bases]
        live = [(sum(widths[:i]), m) for i, m in enumerate(mods)
                if hasattr(m, "lora_delta") and not m.merged]
        if not live:
            return ops.linear(x, w)
        if len({(m.scaling, m.drop_p()) for _, m in live}) == 1:
            # one autograd node on the in-tree kernels: masks applied while x is loaded, updates accumulated
            # in place onto their columns, the backward reading the gradient's column blocks where they are
            m0 = live[0][1]
            y = ops.lora_projection(x, w, [m.adapter_entry(off) for off, m in live], m0.scaling, m0.drop_p())
            if y is not None:
                return y
        qkv = ops.linear(x, w)
        deltas = [m.lora_delta(x) if hasattr(m, "lora_delta") else None for m in mods]
        return ops.add_column_blocks(qkv, deltas, widths)

    def forward(self, x, positions, rope, past_kv, key_padding_mask, residual):
        b, s, _ = x.shape
        cos, sin = rope
        w = self.__dict__.get("_w_qkv")
        bases = self._qkv_bases() if w is not None else ()
        if w is not None and group_intact(w, bases, x):
            qkv = self._qkv_with_adapters(x, w, bases)
            q, k, v = ops.rope_qkv_split(qkv, positions, cos, sin,
                                         self.num_heads, self.num_kv_heads, self.head_dim)
        else:
            q = self.q_proj(x).view(b, s, self.num_heads, self.head_dim)
            k = self.k_proj(x).view(b, s, self.num_kv_heads, self.head_dim)
            v = self.v_proj(x).view(b, s, self.num_kv_heads, self.head_dim)
            q = ops.apply_rope(q, positions, cos, sin)
            k = ops.apply_rope(k, positions, cos, sin)
        # caches are exposed in the [B, H, L, D] layout callers expect
        k_bhld, v_bhld = k.transpose(1, 2), v.transpose(1, 2)
        if past_kv is not None:
            k_bhld = torch.cat([past_kv[0], k_bhld], dim=2)
            v_bhld = torch.cat([past_kv[1], v_bhld], dim=2)
        k_all, v_all = k_bhld.transpose(1, 2), v_bhld.transpose(1, 2)
        if self.num_kv_heads != self.num_heads:
            rep = self.num_heads // self.num_kv_heads
            k_all = k_all.repeat_interleave(rep, dim=2)
            v_all = v_all.repeat_interleave(rep, dim=2)
        o = ops.attention(q, k_all.contiguous(), v_all.contiguous(), causal=True,
                          q_offset=k_all.shape[1] - s, key_padding_mask=key_padding_mask)
        out = self.o_proj(o.reshape(b, s, -1), residual=residual)
        return out, (k_bhld, v_bhld)


class LlamaMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        h, i = config.hidden_size, config.intermediate_size
        self.gate_proj = Linear(h, i, bias=False)
        self.up_proj = Linear(h, i, bias=False)
        self.down_proj = Linear(i, h, bias=False)

    def group_frozen(self) -> bool:
        """``group_frozen_weights`` over gate / up: their stacked rows are the ``[gate | up]`` layout the
        activation reads, so the concatenation disappears from the forward as well."""
        self._w_gate_up = group_frozen_weights((self.gate_proj, self.up_proj))
        return self._w_gate_up is not None

    def forward(self, x, residual=None):
        w = self.__dict__.get("_w_gate_up")
        if w is not None and group_intact(w, (self.gate_proj, self.up_proj), x):
            gate_up = ops.linear(x, w)
        else:
            gate_up = torch.cat([self.gate_proj(x), self.up_proj(x)], dim=-1)
        return self.down_proj(ops.swiglu(gate_up), residual=residual)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = LlamaAttention(config)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, x, positions, rope, past_kv, key_padding_mask):
        h, kv = self.self_attn(self.input_layernorm(x), positions, rope, past_kv,
                               key_padding_mask, residual=x)
        out = self.mlp(self.post_attention_layernorm(h), residual=h)
        return out, kv


class LlamaModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList(LlamaDecoderLayer(config)
                                    for _ in range(config.num_hidden_layers))
        self.norm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self._rope_cache = None
        self.gradient_checkpointing = False

    def rope(self, device):
        c = self._rope_cache
        if c is None or c[0].device != device:
            n = max(self.config.max_position_embeddings, 4096)
            c = ops.rope_tables(n, self.config.head_dim, self.config.rope_theta, device)
            self._rope_cache = c
        return c

    def forward(self, input_ids=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, use_cache=False,
                output_hidden_states=False, return_dict=True, right_padded: bool = False):
        """``right_padded``: the caller vouches that ``attention_mask`` only marks padding at the END of
        each sequence.  Without a cache, causal attention then needs no key mask at all -- a real
        position never looks at a later one -- so batches of unequal length stay on the fused
        attention kernels (rows at padded positions hold meaningless values; their labels are
        ignored and they receive no gradient)."""
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("pass exactly one of input_ids / inputs_embeds")
        x = self.embed_tokens(input_ids) if inputs_embeds is None else inputs_embeds
        b, s, _ = x.shape
        past_len = 0 if not past_key_values else past_key_values[0][0].shape[2]
        if position_ids is None:
            position_ids = torch.arange(past_len, past_len + s, device=x.device)[None].expand(b, s)
        elif position_ids.ndim == 1:
            position_ids = position_ids[None].expand(b, s)
        pad_mask = None
        if attention_mask is not None and not (right_padded and past_len == 0):
            am = attention_mask.bool()
            if not bool(am.all()):
                pad_mask = am
        rope = self.rope(x.device)
        all_h = [x] if output_hidden_states else None
        new_kv = [] if use_cache else None
        for i, layer in enumerate(self.layers):
            past = past_key_values[i] if past_key_values else None
            if self.gradient_checkpointing and self.training and torch.is_grad_enabled():
                from torch.utils.checkpoint import checkpoint
                x, kv = checkpoint(layer, x, position_ids, rope, past, pad_mask,
                                   use_reentrant=False)
            else:
                x, kv = layer(x, position_ids, rope, past, pad_mask)
            if use_cache:
                new_kv.append(kv)
            if output_hidden_states and i < len(self.layers) - 1:
                all_h.append(x)
        x = self.norm(x)
        if output_hidden_states:
            all_h.append(x)  # last entry is the post-norm state, as in HF
        return DecoderOutput(x, tuple(new_kv) if use_cache else None,
                             tuple(all_h) if output_hidden_states else None)


def kv_cache_bytes(config, length: int, batch: int = 1, dtype_bytes: int = 2) -> int:
    """Size of a full KV cache (all layers, K and V)."""
    return (2 * config.num_hidden_layers * batch * config.num_key_value_heads
            * length * config.head_dim * dtype_bytes)
