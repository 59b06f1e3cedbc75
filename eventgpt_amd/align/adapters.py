"""Hidden-state adapters: drafter hidden space -> verifier hidden space.

Levels (behaviour and parameter counts of reference
pipeline/adapter_train/hidden_adapter.py:40-1265; state-dict keys are kept so
checkpoints trained there load here):

  L1  ``HiddenStateAdapter``          N x bottleneck (4096->256->4096, zero-init up) + LN   2.1 M
  L2  ``MultiLayerBottleneckAdapter`` 3 bottleneck blocks                                    6.3 M
  L3  ``WideBottleneckAdapter``       bottleneck 1024, inner LN, per-block alpha              16 M
  L4  ``AttentionAdapter``            pre-LN MHA(8 heads) + FFN(2048) + identity-init proj   101 M
  L5  ``EAGLEStyleAdapter``           + learned positions, causal, next-state loss           103 M
  L5F ``FusedEAGLEAdapter``           sigmoid-gated fusion of (h_drafter, h_verifier)        170 M
  B1  = L5 trained verifier->verifier (upper bound), same class
      ``MultiLayerHiddenAdapter``     fuses the last N decoder layers, then a bottleneck

Design: one base class owns loss, checkpoint IO and the type registry; levels
only define their modules and ``forward``.  The transformer block is a single
implementation (packed in-proj like ``nn.MultiheadAttention`` so keys match)
whose attention runs through ``torch.nn.functional.scaled_dot_product_attention``
— head_dim is 512 here, outside the fused HIP kernels' 64/128.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional, Tuple, Type

import torch
import torch.nn.functional as F
from torch import nn

from .losses import alignment_losses


@dataclasses.dataclass
class HiddenAdapterConfig:
    hidden_dim: int = 4096
    bottleneck_dim: int = 256
    num_layers: int = 1
    dropout: float = 0.1
    alpha: float = 0.1
    use_layer_norm: bool = True
    use_gelu: bool = True


_REGISTRY: Dict[str, Type["HiddenAdapterBase"]] = {}


def register(adapter_type: str):
    def deco(cls):
        cls.adapter_type = adapter_type
        _REGISTRY[adapter_type] = cls
        return cls
    return deco


class HiddenAdapterBase(nn.Module):
    """Loss, checkpointing and bookkeeping shared by all adapter levels."""

    adapter_type = "base"
    prediction_weight = 0.0      # > 0 adds the next-position loss (L5 / L5F)
    takes_target_input = False   # L5F consumes the verifier states as a second input

    def config_dict(self) -> Dict[str, Any]:
        raise NotImplementedError

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def align(self, source: torch.Tensor, target: Optional[torch.Tensor] = None,
              attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Uniform call: levels that ignore ``target`` / ``attention_mask`` drop them."""
        return self.forward(source)

    def compute_loss(self, egpt_hidden: torch.Tensor, vl_hidden: torch.Tensor,
                     attention_mask: Optional[torch.Tensor] = None,
                     prediction_weight: Optional[float] = None) -> Dict[str, torch.Tensor]:
        pred = self.align(egpt_hidden, vl_hidden, attention_mask)
        w = self.prediction_weight if prediction_weight is None else prediction_weight
        return alignment_losses(pred, vl_hidden, attention_mask, w)

    # -- checkpoint -----------------------------------------------------------
    def save_checkpoint(self, path: str, optimizer=None, epoch: int = 0,
                        metrics: Optional[dict] = None, extra: Optional[dict] = None) -> None:
        ckpt = {"model_state_dict": self.state_dict(), "adapter_type": self.adapter_type,
                "config": self.config_dict(), "epoch": epoch}
        if optimizer is not None:
            ckpt["optimizer_state_dict"] = optimizer.state_dict()
        if metrics is not None:
            ckpt["metrics"] = metrics
        if extra:
            ckpt.update(extra)
        torch.save(ckpt, path)

    @classmethod
    def from_checkpoint_dict(cls, ckpt: dict) -> "HiddenAdapterBase":
        cfg = ckpt["config"]
        if not isinstance(cfg, dict):
            cfg = {f.name: getattr(cfg, f.name) for f in dataclasses.fields(HiddenAdapterConfig)
                   if hasattr(cfg, f.name)}
        model = cls.from_config_dict(cfg)
        model.load_state_dict(ckpt["model_state_dict"])
        return model

    @classmethod
    def from_config_dict(cls, cfg: dict) -> "HiddenAdapterBase":
        return cls(**cfg)

    @classmethod
    def load_checkpoint(cls, path: str, device: str = "cpu"):
        ckpt = _load_ckpt(path, device)
        return cls.from_checkpoint_dict(ckpt).to(device), ckpt


def _load_ckpt(path: str, device) -> dict:
    try:
        return torch.load(path, map_location=device, weights_only=True)
    except Exception:
        # checkpoints that pickle a config dataclass need the class allow-listed
        import torch.serialization as ts
        with ts.safe_globals([HiddenAdapterConfig]):
            return torch.load(path, map_location=device, weights_only=True)


# --------------------------------------------------------------- building blocks
class BottleneckAdapter(nn.Module):
    """``h + alpha * up(drop(act(down(LN(h)))))`` with a zero-initialised ``up``."""

    def __init__(self, hidden_dim: int = 4096, bottleneck_dim: int = 256, dropout: float = 0.1,
                 alpha: float = 0.1, use_layer_norm: bool = True, use_gelu: bool = True):
        super().__init__()
        self.down_proj = nn.Linear(hidden_dim, bottleneck_dim, bias=False)
        self.up_proj = nn.Linear(bottleneck_dim, hidden_dim, bias=False)
        self.layer_norm = nn.LayerNorm(hidden_dim) if use_layer_norm else nn.Identity()
        self.act = nn.GELU() if use_gelu else nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.alpha_param = nn.Parameter(torch.tensor(float(alpha)))
        nn.init.kaiming_normal_(self.down_proj.weight, mode="fan_in", nonlinearity="linear")
        nn.init.zeros_(self.up_proj.weight)

    def forward(self, h):
        return h + self.alpha_param * self.up_proj(self.dropout(self.act(self.down_proj(
            self.layer_norm(h)))))

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class SelfAttention(nn.Module):
    """Multi-head self-attention with ``nn.MultiheadAttention``'s parameter names."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x, causal: bool = False, key_padding_mask: Optional[torch.Tensor] = None):
        b, s, e = x.shape
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)
        if e // self.num_heads == 512 and qkv.is_cuda:
            from ..ops import differentiable as diff
            if diff.attention_d512_ok(qkv, self.num_heads):
                o = diff.attention_d512(qkv, self.num_heads, causal, key_padding_mask,
                                        self.dropout if self.training else 0.0)
                return self.out_proj(o)
        q, k, v = qkv.view(b, s, 3, self.num_heads, e // self.num_heads).permute(2, 0, 3, 1, 4)
        mask = None
        if key_padding_mask is not None:      # True = padded key
            mask = ~key_padding_mask[:, None, None, :].expand(b, 1, s, s)
            if causal:
                mask = mask & torch.ones(s, s, dtype=torch.bool, device=x.device).tril()
        o = F.scaled_dot_product_attention(
            q, k, v, attn_mask=mask, dropout_p=self.dropout if self.training else 0.0,
            is_causal=causal and mask is None)
        return self.out_proj(o.transpose(1, 2).reshape(b, s, e))


def _ffn(hidden_dim: int, ffn_dim: int, dropout: float) -> nn.Sequential:
    return nn.Sequential(nn.Linear(hidden_dim, ffn_dim), nn.GELU(), nn.Dropout(dropout),
                         nn.Linear(ffn_dim, hidden_dim), nn.Dropout(dropout))


def _blocks(hidden_dim, num_heads, ffn_dim, num_layers, dropout) -> nn.ModuleList:
    return nn.ModuleList(nn.ModuleDict({
        "attn_norm": nn.LayerNorm(hidden_dim),
        "attn": SelfAttention(hidden_dim, num_heads, dropout),
        "ffn_norm": nn.LayerNorm(hidden_dim),
        "ffn": _ffn(hidden_dim, ffn_dim, dropout)}) for _ in range(num_layers))


def _run_blocks(layers, x, causal, attention_mask):
    kpm = None if attention_mask is None else (attention_mask == 0)
    for layer in layers:
        x = x + layer["attn"](layer["attn_norm"](x), causal=causal, key_padding_mask=kpm)
        x = x + layer["ffn"](layer["ffn_norm"](x))
    return x


# ------------------------------------------------------------------------ levels
@register("L1_Bottleneck")
class HiddenStateAdapter(HiddenAdapterBase):
    def __init__(self, config: Optional[HiddenAdapterConfig] = None, **kw):
        super().__init__()
        self.config = config or HiddenAdapterConfig(**kw)
        c = self.config
        self.layers = nn.ModuleList(
            BottleneckAdapter(c.hidden_dim, c.bottleneck_dim, c.dropout, c.alpha,
                              c.use_layer_norm, c.use_gelu) for _ in range(c.num_layers))
        self.final_norm = nn.LayerNorm(c.hidden_dim)

    def forward(self, hidden_states):
        x = hidden_states
        for layer in self.layers:
            x = layer(x)
        return self.final_norm(x)

    def config_dict(self):
        return dataclasses.asdict(self.config)

    @classmethod
    def from_config_dict(cls, cfg):
        return cls(HiddenAdapterConfig(**cfg))


@register("L2_MultiLayerBottleneck")
class MultiLayerBottleneckAdapter(HiddenAdapterBase):
    def __init__(self, hidden_dim: int = 4096, bottleneck_dim: int = 256, num_blocks: int = 3,
                 dropout: float = 0.1, alpha: float = 0.1):
        super().__init__()
        self.hidden_dim, self.bottleneck_dim, self.num_blocks = hidden_dim, bottleneck_dim, num_blocks
        self.blocks = nn.ModuleList(BottleneckAdapter(hidden_dim, bottleneck_dim, dropout, alpha)
                                    for _ in range(num_blocks))
        self.final_norm = nn.LayerNorm(hidden_dim)

    def forward(self, hidden_states):
        x = hidden_states
        for blk in self.blocks:
            x = blk(x)
        return self.final_norm(x)

    def config_dict(self):
        return {"hidden_dim": self.hidden_dim, "bottleneck_dim": self.bottleneck_dim,
                "num_blocks": self.num_blocks}


@register("L3_WideBottleneck")
class WideBottleneckAdapter(HiddenAdapterBase):
    def __init__(self, hidden_dim: int = 4096, bottleneck_dim: int = 1024, num_blocks: int = 2,
                 dropout: float = 0.1, alpha: float = 0.1):
        super().__init__()
        self.hidden_dim, self.bottleneck_dim, self.num_blocks = hidden_dim, bottleneck_dim, num_blocks
        self.input_norm = nn.LayerNorm(hidden_dim)
        self.blocks = nn.ModuleList(nn.Sequential(
            nn.Linear(hidden_dim, bottleneck_dim, bias=False), nn.LayerNorm(bottleneck_dim),
            nn.GELU(), nn.Dropout(dropout), nn.Linear(bottleneck_dim, hidden_dim, bias=False),
            nn.Dropout(dropout)) for _ in range(num_blocks))
        self.alphas = nn.ParameterList(nn.Parameter(torch.tensor(float(alpha)))
                                       for _ in range(num_blocks))
        self.final_norm = nn.LayerNorm(hidden_dim)
        for blk in self.blocks:
            nn.init.kaiming_normal_(blk[0].weight, mode="fan_in")
            nn.init.zeros_(blk[4].weight)

    def forward(self, hidden_states):
        x = self.input_norm(hidden_states)
        for blk, a in zip(self.blocks, self.alphas):
            x = x + a * blk(x)
        return self.final_norm(x)

    def config_dict(self):
        return {"hidden_dim": self.hidden_dim, "bottleneck_dim": self.bottleneck_dim,
                "num_blocks": self.num_blocks}


@register("L4_Attention")
class AttentionAdapter(HiddenAdapterBase):
    def __init__(self, hidden_dim: int = 4096, num_heads: int = 8, ffn_dim: int = 2048,
                 num_layers: int = 1, dropout: float = 0.1):
        super().__init__()
        self.hidden_dim, self.num_heads, self.ffn_dim, self.num_layers = \
            hidden_dim, num_heads, ffn_dim, num_layers
        self.layers = _blocks(hidden_dim, num_heads, ffn_dim, num_layers, dropout)
        self.output_proj = nn.Sequential(nn.LayerNorm(hidden_dim), nn.Linear(hidden_dim, hidden_dim))
        self.alpha = nn.Parameter(torch.tensor(0.1))
        nn.init.eye_(self.output_proj[1].weight)
        nn.init.zeros_(self.output_proj[1].bias)

    def forward(self, hidden_states, attention_mask=None):
        x = _run_blocks(self.layers, hidden_states, False, attention_mask)
        return hidden_states + self.alpha * (self.output_proj(x) - hidden_states)

    def align(self, source, target=None, attention_mask=None):
        return self.forward(source, attention_mask)

    def config_dict(self):
        return {"hidden_dim": self.hidden_dim, "num_heads": self.num_heads,
                "ffn_dim": self.ffn_dim, "num_layers": self.num_layers}


@register("L5_EAGLE")
class EAGLEStyleAdapter(HiddenAdapterBase):
    """Causal block over position-tagged states; trained on alignment + next state."""

    prediction_weight = 0.5

    def __init__(self, hidden_dim: int = 4096, num_heads: int = 8, ffn_dim: int = 2048,
                 num_layers: int = 1, dropout: float = 0.1, max_seq_len: int = 512,
                 vocab_size: int = 32000, use_token_embed: bool = False):
        super().__init__()
        self.hidden_dim, self.num_heads, self.ffn_dim, self.num_layers = \
            hidden_dim, num_heads, ffn_dim, num_layers
        self.use_token_embed = use_token_embed
        if use_token_embed:
            self.token_embed = nn.Embedding(vocab_size, hidden_dim)
            self.token_fusion = nn.Linear(hidden_dim * 2, hidden_dim)
        self.input_norm = nn.LayerNorm(hidden_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, max_seq_len, hidden_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.layers = _blocks(hidden_dim, num_heads, ffn_dim, num_layers, dropout)
        self.output_norm = nn.LayerNorm(hidden_dim)
        self.output_proj = nn.Linear(hidden_dim, hidden_dim)
        self.alpha = nn.Parameter(torch.tensor(0.1))
        nn.init.eye_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, hidden_states, prev_tokens=None, attention_mask=None):
        x = hidden_states
        if self.use_token_embed and prev_tokens is not None:
            x = self.token_fusion(torch.cat([x, self.token_embed(prev_tokens)], dim=-1))
        x = self.input_norm(x) + self.pos_embed[:, :x.shape[1]]
        x = _run_blocks(self.layers, x, True, attention_mask)
        out = self.output_proj(self.output_norm(x))
        return hidden_states + self.alpha * (out - hidden_states)

    def align(self, source, target=None, attention_mask=None):
        return self.forward(source, attention_mask=attention_mask)

    @torch.no_grad()
    def speculative_decode(self, initial_hidden: torch.Tensor, num_draft_tokens: int = 5):
        """Roll the adapter forward on its own output: ``[B,1,H]`` -> ``[B,n,H]``."""
        states = [initial_hidden]
        cur = initial_hidden
        for _ in range(num_draft_tokens - 1):
            cur = self.forward(cur)[:, -1:]
            states.append(cur)
        return torch.cat(states, dim=1)

    def config_dict(self):
        return {"hidden_dim": self.hidden_dim, "num_heads": self.num_heads,
                "ffn_dim": self.ffn_dim, "num_layers": self.num_layers,
                "use_token_embed": self.use_token_embed}


@register("L5F_FusedEAGLE")
class FusedEAGLEAdapter(HiddenAdapterBase):
    """Gated fusion of drafter and verifier states, residual from the verifier's."""

    prediction_weight = 0.5
    takes_target_input = True

    def __init__(self, hidden_dim: int = 4096, num_heads: int = 8, ffn_dim: int = 2048,
                 num_layers: int = 1, dropout: float = 0.1, max_seq_len: int = 512):
        super().__init__()
        self.hidden_dim, self.num_heads, self.ffn_dim, self.num_layers = \
            hidden_dim, num_heads, ffn_dim, num_layers
        self.fusion_proj = nn.Linear(hidden_dim * 2, hidden_dim)
        self.fusion_norm = nn.LayerNorm(hidden_dim)
        self.gate = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.Sigmoid())
        self.pos_embed = nn.Parameter(torch.zeros(1, max_seq_len, hidden_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.layers = _blocks(hidden_dim, num_heads, ffn_dim, num_layers, dropout)
        self.output_norm = nn.LayerNorm(hidden_dim)
        self.output_proj = nn.Linear(hidden_dim, hidden_dim)
        self.alpha = nn.Parameter(torch.tensor(0.1))
        nn.init.eye_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)
        nn.init.xavier_uniform_(self.fusion_proj.weight)
        nn.init.zeros_(self.fusion_proj.bias)

    def forward(self, egpt_hidden, vl_hidden, attention_mask=None):
        both = torch.cat([egpt_hidden, vl_hidden], dim=-1)
        g = self.gate(both)
        x = self.fusion_norm(g * egpt_hidden + (1 - g) * vl_hidden + self.fusion_proj(both))
        x = x + self.pos_embed[:, :x.shape[1]]
        x = _run_blocks(self.layers, x, True, attention_mask)
        out = self.output_proj(self.output_norm(x))
        return vl_hidden + self.alpha * (out - vl_hidden)

    def align(self, source, target=None, attention_mask=None):
        if target is None:
            raise ValueError("L5F needs the verifier hidden states as its second input")
        return self.forward(source, target, attention_mask)

    def config_dict(self):
        return {"hidden_dim": self.hidden_dim, "num_heads": self.num_heads,
                "ffn_dim": self.ffn_dim, "num_layers": self.num_layers}


class MultiLayerHiddenAdapter(nn.Module):
    """Fuse the last ``num_input_layers`` decoder layers, then one bottleneck."""

    def __init__(self, hidden_dim: int = 4096, bottleneck_dim: int = 256,
                 num_input_layers: int = 4, dropout: float = 0.1):
        super().__init__()
        self.hidden_dim, self.num_input_layers = hidden_dim, num_input_layers
        self.layer_fusion = nn.Sequential(nn.Linear(hidden_dim * num_input_layers, hidden_dim),
                                          nn.LayerNorm(hidden_dim), nn.GELU(), nn.Dropout(dropout))
        self.adapter = BottleneckAdapter(hidden_dim, bottleneck_dim, dropout)

    def forward(self, layer_hidden_states):  # [B, L, S, H]
        b, l, s, h = layer_hidden_states.shape
        return self.adapter(self.layer_fusion(
            layer_hidden_states.permute(0, 2, 1, 3).reshape(b, s, l * h)))


# ---------------------------------------------------------------------- factories
LEVEL_NAMES = {1: "L1_Bottleneck", 2: "L2_MultiLayerBottleneck", 3: "L3_WideBottleneck",
               4: "L4_Attention", 5: "L5_EAGLE", 6: "L5F_FusedEAGLE"}


def create_hidden_adapter(hidden_dim: int = 4096, bottleneck_dim: int = 256, num_layers: int = 1,
                          alpha: float = 0.1) -> HiddenStateAdapter:
    return HiddenStateAdapter(HiddenAdapterConfig(hidden_dim=hidden_dim, bottleneck_dim=bottleneck_dim,
                                                  num_layers=num_layers, alpha=alpha))


def create_adapter(level: int = 1, hidden_dim: int = 4096, verbose: bool = False,
                   **kw) -> HiddenAdapterBase:
    """Adapter of complexity ``level`` in 1..6 (6 = L5F); unknown kwargs are ignored
    per level exactly like the reference factory (hidden_adapter.py:1308-1423)."""
    g = kw.get
    if level == 1:
        m = create_hidden_adapter(hidden_dim, g("bottleneck_dim", 256), g("num_layers", 1),
                                  g("alpha", 0.1))
    elif level == 2:
        m = MultiLayerBottleneckAdapter(hidden_dim, g("bottleneck_dim", 256), g("num_blocks", 3),
                                        g("dropout", 0.1), g("alpha", 0.1))
    elif level == 3:
        m = WideBottleneckAdapter(hidden_dim, g("bottleneck_dim", 1024), g("num_blocks", 2),
                                  g("dropout", 0.1), g("alpha", 0.1))
    elif level == 4:
        m = AttentionAdapter(hidden_dim, g("num_heads", 8), g("ffn_dim", 2048),
                             g("num_layers", 1), g("dropout", 0.1))
    elif level == 5:
        m = EAGLEStyleAdapter(hidden_dim, g("num_heads", 8), g("ffn_dim", 2048),
                              g("num_layers", 1), g("dropout", 0.1),
                              use_token_embed=g("use_token_embed", False))
    elif level == 6:
        m = FusedEAGLEAdapter(hidden_dim, g("num_heads", 8), g("ffn_dim", 2048),
                              g("num_layers", 1), g("dropout", 0.1))
    else:
        raise ValueError(f"Invalid level {level}. Must be 1-6.")
    if verbose:
        n = m.get_num_parameters()
        print(f"Created {LEVEL_NAMES[level]}: {n:,} parameters ({n * 4 / 2**20:.1f} MB fp32)")
    return m


def load_any_adapter(path: str, device: str = "cpu") -> Tuple[HiddenAdapterBase, dict]:
    """Load any adapter checkpoint, dispatching on its ``adapter_type`` (L1 files
    written by the reference carry none)."""
    ckpt = _load_ckpt(path, device)
    cls = _REGISTRY.get(ckpt.get("adapter_type", "L1_Bottleneck"), HiddenStateAdapter)
    return cls.from_checkpoint_dict(ckpt).to(device), ckpt


def self_test(hidden_dim: int = 4096, batch: int = 4, seq: int = 50, device: str = "cpu", iters: int = 10,
              out=print) -> list:
    """Build every adapter level on random ``[batch, seq, hidden]`` inputs, time ``iters``
    forwards, take one loss, and print the parameter table (the reference module's
    ``__main__`` smoke check, hidden_adapter.py:1454-1522).  Returns the table rows."""
    import time
    rows = []
    g = torch.Generator().manual_seed(0)
    e = torch.randn(batch, seq, hidden_dim, generator=g).to(device)
    v = (e + 0.1 * torch.randn(batch, seq, hidden_dim, generator=g).to(device))
    mask = torch.ones(batch, seq, device=device)
    for level in range(1, 7):
        m = create_adapter(level, hidden_dim).to(device).eval()
        with torch.no_grad():
            m.align(e, v, mask)
            if device != "cpu":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                pred = m.align(e, v, mask)
            if device != "cpu":
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / iters
        loss = m.compute_loss(e, v, mask)
        assert pred.shape == e.shape and torch.isfinite(loss["total_loss"])
        rows.append({"level": LEVEL_NAMES[level], "parameters": m.get_num_parameters(), "forward_ms": ms,
                     "loss": float(loss["total_loss"])})
        out(f"{LEVEL_NAMES[level]:>4s}  {rows[-1]['parameters'] / 1e6:8.1f} M params  {ms:8.2f} ms/forward  "
            f"loss {rows[-1]['loss']:.4f}")
    return rows


if __name__ == "__main__":
    import sys
    self_test(device="cuda" if torch.cuda.is_available() and "--cpu" not in sys.argv else "cpu")
