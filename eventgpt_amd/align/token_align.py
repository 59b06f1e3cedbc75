"""Token-level alignment: make a drafter's token stream acceptable to the verifier.

Modules (reference counterparts in brackets, all under feasible/token_alignment/):

* ``TokenAdapter`` — draft ids -> causal encoder -> verifier-vocabulary logits; the
  deployed 45 M variant and the library variant with a learnable ``token_bias`` and an
  entropy auxiliary loss are one class here [train_and_evaluate.py:243-300,
  token_adapter.py:49-254].  State-dict keys match ``nn.TransformerEncoder`` so the
  reference's ``best_model.pt`` loads.
* ``EAGLEFusionModule`` — (draft hidden ++ previous-token embedding) -> causal fusion
  layers -> logits (+ Medusa heads, feature head) [eagle_fusion.py:39-473].
* ``TokenProjectionModule``, ``LogitAlignmentModule``, ``SequenceDistillationModule``
  [token_projection.py:39-189, logit_alignment.py:39-153, sequence_distillation.py:39-160].
* ``TokenAlignmentDataset`` (json / directory IO), ``AlignmentTrainer`` (AdamW, clip,
  early stopping) [base.py:193-462], acceptance metrics [train_and_evaluate.py:303-383],
  SD simulation [evaluate_speculative.py:42-206].

One causal encoder implementation serves every module; attention goes through
``scaled_dot_product_attention`` (flash path on ROCm).  Checkpoints hold plain dicts
(state dict + config dict), never pickled config objects.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

PAD_ID = 0


# ------------------------------------------------------------------ building blocks
class _SelfAttn(nn.Module):
    """Packed-QKV attention with ``nn.MultiheadAttention``'s parameter names."""

    def __init__(self, dim: int, heads: int, dropout: float = 0.0):
        super().__init__()
        if dim % heads:
            raise ValueError("embed dim must divide by heads")
        self.heads, self.dropout = heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = nn.Linear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, x, causal: bool = True, key_padding_mask: Optional[torch.Tensor] = None):
        b, s, d = x.shape
        q, k, v = F.linear(x, self.in_proj_weight, self.in_proj_bias).view(b, s, 3, self.heads, -1).unbind(2)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))
        mask = None
        if key_padding_mask is not None:                 # True = padded key
            mask = ~key_padding_mask[:, None, None, :].expand(b, 1, s, s)
            if causal:
                mask = mask & torch.ones(s, s, dtype=torch.bool, device=x.device).tril()
            # a fully masked row (padded query) would produce NaN: let it see itself
            mask = mask | torch.eye(s, dtype=torch.bool, device=x.device)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=causal and mask is None,
                                           dropout_p=self.dropout if self.training else 0.0)
        return self.out_proj(o.transpose(1, 2).reshape(b, s, d))


class EncoderLayer(nn.Module):
    """Post-norm (default) or pre-norm transformer block, GELU FFN."""

    def __init__(self, dim: int, heads: int, ffn_dim: int, dropout: float = 0.1, pre_norm: bool = False):
        super().__init__()
        self.self_attn = _SelfAttn(dim, heads, dropout)
        self.linear1, self.linear2 = nn.Linear(dim, ffn_dim), nn.Linear(ffn_dim, dim)
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.drop = nn.Dropout(dropout)
        self.pre_norm = pre_norm

    def _ffn(self, x):
        return self.drop(self.linear2(self.drop(F.gelu(self.linear1(x)))))

    def forward(self, x, causal=True, key_padding_mask=None):
        if self.pre_norm:
            x = x + self.drop(self.self_attn(self.norm1(x), causal, key_padding_mask))
            return x + self._ffn(self.norm2(x))
        x = self.norm1(x + self.drop(self.self_attn(x, causal, key_padding_mask)))
        return self.norm2(x + self._ffn(x))


class CausalEncoder(nn.Module):
    def __init__(self, dim, heads, ffn_dim, num_layers, dropout=0.1, pre_norm=False):
        super().__init__()
        self.layers = nn.ModuleList(EncoderLayer(dim, heads, ffn_dim, dropout, pre_norm) for _ in range(num_layers))

    def forward(self, x, key_padding_mask=None, causal=True):
        for layer in self.layers:
            x = layer(x, causal, key_padding_mask)
        return x


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers=2, dropout=0.1, use_layer_norm=True):
        super().__init__()
        mods, d = [], input_dim
        for _ in range(num_layers - 1):
            mods.append(nn.Linear(d, hidden_dim))
            if use_layer_norm:
                mods.append(nn.LayerNorm(hidden_dim))
            mods += [nn.GELU(), nn.Dropout(dropout)]
            d = hidden_dim
        mods.append(nn.Linear(d, output_dim))
        self.mlp = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp(x)


def _init_normal(module: nn.Module, std: float = 0.02) -> None:
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, std=std)


def masked_token_accuracy(logits: torch.Tensor, labels: torch.Tensor, k: int = 1) -> Tuple[float, int, int]:
    mask = labels != PAD_ID
    if k == 1:
        hit = logits.argmax(-1) == labels
    else:
        hit = (logits.topk(min(k, logits.shape[-1]), -1).indices == labels.unsqueeze(-1)).any(-1)
    n = int(mask.sum())
    c = int((hit & mask).sum())
    return (c / n if n else 0.0), c, n


@dataclasses.dataclass
class TokenAlignmentConfig:
    draft_vocab_size: int = 32000
    target_vocab_size: int = 32000
    draft_hidden_size: int = 4096
    target_hidden_size: int = 4096
    hidden_dim: int = 2048
    num_layers: int = 2
    dropout: float = 0.1
    temperature: float = 1.0
    use_layer_norm: bool = True
    use_residual: bool = False

    def to_dict(self):
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, d):
        names = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})


class BaseTokenAlignmentModule(nn.Module):
    config_class = TokenAlignmentConfig

    def __init__(self, config):
        super().__init__()
        self.config = config

    @staticmethod
    def compute_kl_loss(draft_logits, target_logits, temperature: float = 1.0):
        lp = F.log_softmax(draft_logits / temperature, -1)
        return F.kl_div(lp, F.softmax(target_logits / temperature, -1), reduction="batchmean") * temperature ** 2

    @staticmethod
    def compute_ce_loss(logits, targets, ignore_index: int = -100):
        return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1), ignore_index=ignore_index)

    @staticmethod
    def compute_mse_loss(predicted, target):
        return F.mse_loss(predicted, target)

    @staticmethod
    def compute_acceptance_rate(draft_tokens, target_tokens, pad_token_id: int = PAD_ID):
        mask = (target_tokens != pad_token_id) & (draft_tokens != pad_token_id)
        n = int(mask.sum())
        c = int(((draft_tokens == target_tokens) & mask).sum())
        return (c / n if n else 0.0), c, n

    def save(self, path: str, **extra) -> None:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({"model_state_dict": self.state_dict(), "config": self.config.to_dict(),
                    "module": type(self).__name__, **extra}, path)

    @classmethod
    def load(cls, path: str, device="cpu"):
        ck = torch.load(path, map_location=device, weights_only=True)
        model = cls(cls.config_class.from_dict(ck["config"]))
        model.load_state_dict(ck["model_state_dict"])
        return model.to(device)


# -------------------------------------------------------------------- TokenAdapter
@dataclasses.dataclass
class TokenAdapterConfig(TokenAlignmentConfig):
    draft_vocab_size: int = 32010
    target_vocab_size: int = 32010
    embed_dim: int = 512
    num_layers: int = 4
    num_heads: int = 8
    ffn_dim: int = 2048
    max_context: int = 128
    label_smoothing: float = 0.1
    use_token_bias: bool = False
    use_auxiliary_loss: bool = False


class TokenAdapter(BaseTokenAlignmentModule):
    config_class = TokenAdapterConfig

    def __init__(self, config: Optional[TokenAdapterConfig] = None, **kw):
        super().__init__(config or TokenAdapterConfig(**kw))
        c = self.config
        self.embedding = nn.Embedding(c.draft_vocab_size, c.embed_dim)
        self.position_embedding = nn.Embedding(c.max_context, c.embed_dim)
        self.transformer = CausalEncoder(c.embed_dim, c.num_heads, c.ffn_dim, c.num_layers, c.dropout)
        self.output_proj = nn.Sequential(nn.LayerNorm(c.embed_dim), nn.Linear(c.embed_dim, c.target_vocab_size))
        if c.use_token_bias:
            self.token_bias = nn.Parameter(torch.zeros(c.target_vocab_size))
        _init_normal(self)

    def forward(self, draft_tokens: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = min(draft_tokens.shape[1], self.config.max_context)
        tok = draft_tokens[:, :n]
        x = self.embedding(tok) + self.position_embedding(torch.arange(n, device=tok.device))[None]
        pad = (attention_mask[:, :n] == 0) if attention_mask is not None else None
        logits = self.output_proj(self.transformer(x, pad))
        return logits + self.token_bias if self.config.use_token_bias else logits

    def compute_loss(self, draft_tokens, target_tokens, attention_mask=None) -> Dict[str, torch.Tensor]:
        """Position t of the draft predicts position t+1 of the target."""
        am = attention_mask[:, :-1] if attention_mask is not None else None
        logits = self(draft_tokens[:, :-1], am)
        labels = target_tokens[:, 1:1 + logits.shape[1]]
        logits = logits[:, :labels.shape[1]]
        ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=PAD_ID,
                             label_smoothing=self.config.label_smoothing)
        aux = logits.new_zeros(())
        if self.config.use_auxiliary_loss:       # minus 0.01 x entropy: a push towards confident predictions
            lp = F.log_softmax(logits.float(), -1)
            aux = (lp.exp() * lp).sum(-1).mean() * 0.01
        with torch.no_grad():
            acc, _, _ = masked_token_accuracy(logits, labels)
            top5, _, _ = masked_token_accuracy(logits, labels, 5)
        return {"total_loss": ce + aux, "ce_loss": ce, "aux_loss": aux, "accuracy": torch.tensor(acc),
                "top5_accuracy": torch.tensor(top5)}

    @classmethod
    def load(cls, path: str, device="cpu") -> "TokenAdapter":
        """Own checkpoints, or the reference's ``{model_state_dict, epoch, val_acc}`` whose
        geometry is read back from the tensor shapes."""
        ck = torch.load(path, map_location=device, weights_only=True)
        sd = ck["model_state_dict"]
        if isinstance(ck.get("config"), dict):
            cfg = TokenAdapterConfig.from_dict(ck["config"])
        else:
            vocab, dim = sd["embedding.weight"].shape
            layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.layers."))
            cfg = TokenAdapterConfig(draft_vocab_size=vocab, target_vocab_size=sd["output_proj.1.weight"].shape[0],
                                     embed_dim=dim, num_layers=layers, num_heads=8,
                                     ffn_dim=sd["transformer.layers.0.linear1.weight"].shape[0],
                                     max_context=sd["position_embedding.weight"].shape[0],
                                     use_token_bias="token_bias" in sd)
        model = cls(cfg)
        model.load_state_dict(sd)
        return model.to(device).eval()


class TokenAdapterForSD:
    """Maps a drafter's token ids to verifier ids for speculative verification
    (reference: feasible/egpt_prefill_only/prefill_then_verify.py:50-103): the adapter
    sees the draft so far and its prediction at position t replaces draft token t+1;
    the first token passes through."""

    def __init__(self, adapter: Optional[TokenAdapter] = None, id_map: Optional[torch.Tensor] = None):
        self.adapter, self.id_map = adapter, id_map

    @torch.no_grad()
    def __call__(self, draft: torch.Tensor) -> torch.Tensor:
        ids = draft.long()
        if self.id_map is not None:
            ids = self.id_map.to(ids.device)[ids.clamp(0, self.id_map.numel() - 1)]
        if self.adapter is None or ids.numel() < 2:
            return ids.to(draft.dtype)
        pred = self.adapter(ids[None, :-1])[0].argmax(-1)
        out = ids.clone()
        out[1:1 + pred.shape[0]] = pred
        return out.to(draft.dtype)


# -------------------------------------------------------------------- EAGLE fusion
@dataclasses.dataclass
class EAGLEFusionConfig(TokenAlignmentConfig):
    fusion_type: str = "eagle"
    num_fusion_layers: int = 2
    num_attention_heads: int = 8
    use_rotary_embeddings: bool = True
    num_speculation_heads: int = 1
    max_speculation_depth: int = 5
    use_kl_loss: bool = True
    kl_weight: float = 0.5
    ce_weight: float = 1.0
    feature_matching_weight: float = 0.1
    max_position: int = 512


class EAGLEFusionLayer(nn.Module):
    def __init__(self, config: EAGLEFusionConfig, in_dim: int):
        super().__init__()
        self.input_proj = nn.Linear(in_dim + config.target_hidden_size, config.hidden_dim)
        self.block = EncoderLayer(config.hidden_dim, config.num_attention_heads, config.hidden_dim * 4,
                                  config.dropout, pre_norm=True)

    def forward(self, stream, token_emb, attention_mask=None):
        x = self.input_proj(torch.cat([stream, token_emb], -1))
        pad = (attention_mask == 0) if attention_mask is not None else None
        return self.block(x, True, pad)


class EAGLEFusionModule(BaseTokenAlignmentModule):
    config_class = EAGLEFusionConfig

    def __init__(self, config: Optional[EAGLEFusionConfig] = None, **kw):
        super().__init__(config or EAGLEFusionConfig(**kw))
        c = self.config
        self.token_embedding = nn.Embedding(c.target_vocab_size, c.target_hidden_size)
        # layer 0 consumes the drafter's hidden state, deeper layers the fused stream
        self.fusion_layers = nn.ModuleList(
            EAGLEFusionLayer(c, c.draft_hidden_size if i == 0 else c.hidden_dim) for i in range(c.num_fusion_layers))
        self.output_proj = nn.Sequential(nn.LayerNorm(c.hidden_dim), nn.Linear(c.hidden_dim, c.target_vocab_size))
        self.speculation_heads = nn.ModuleList(
            nn.Linear(c.hidden_dim, c.target_vocab_size) for _ in range(c.num_speculation_heads - 1)) \
            if c.num_speculation_heads > 1 else None
        self.feature_proj = nn.Linear(c.hidden_dim, c.target_hidden_size)
        _init_normal(self)

    def forward(self, draft_hidden, input_tokens=None, input_embeds=None, attention_mask=None) -> Dict:
        b, s, _ = draft_hidden.shape
        if input_embeds is not None:
            emb = input_embeds
        elif input_tokens is not None:
            emb = self.token_embedding(input_tokens)
        else:
            emb = draft_hidden.new_zeros(b, s, self.config.target_hidden_size)
        if emb.shape[1] < s:
            emb = F.pad(emb, (0, 0, 0, s - emb.shape[1]))
        emb = emb[:, :s].to(draft_hidden.dtype)
        x = draft_hidden
        for layer in self.fusion_layers:
            x = layer(x, emb, attention_mask)
        out = {"logits": self.output_proj(x), "features": self.feature_proj(x)}
        if self.speculation_heads is not None:
            out["speculation_logits"] = torch.stack([h(x) for h in self.speculation_heads], 1)
        return out

    def compute_loss(self, draft_hidden, target_tokens, target_hidden=None, target_logits=None,
                     attention_mask=None) -> Dict[str, torch.Tensor]:
        c = self.config
        am = attention_mask[:, :-1] if attention_mask is not None else None
        out = self(draft_hidden[:, :-1], input_tokens=target_tokens[:, :-1], attention_mask=am)
        logits, labels = out["logits"], target_tokens[:, 1:]
        ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=PAD_ID,
                             label_smoothing=0.1)
        kl = logits.new_zeros(())
        if target_logits is not None and c.use_kl_loss:
            kl = self.compute_kl_loss(logits, target_logits[:, 1:], c.temperature)
        feat = logits.new_zeros(())
        if target_hidden is not None:
            feat = F.mse_loss(out["features"], target_hidden[:, 1:])
        spec = logits.new_zeros(())
        if self.speculation_heads is not None:       # head k predicts the token k+1 steps further
            for k in range(out["speculation_logits"].shape[1]):
                lg = out["speculation_logits"][:, k, :-(k + 1)]
                lb = target_tokens[:, 2 + k:]
                if lb.shape[1]:
                    spec = spec + F.cross_entropy(lg.reshape(-1, lg.shape[-1]), lb.reshape(-1), ignore_index=PAD_ID)
        total = c.ce_weight * ce + c.kl_weight * kl + c.feature_matching_weight * feat + 0.5 * spec
        with torch.no_grad():
            acc, _, _ = masked_token_accuracy(logits, labels)
        return {"total_loss": total, "ce_loss": ce, "kl_loss": kl, "feature_loss": feat, "spec_loss": spec,
                "accuracy": torch.tensor(acc)}

    @torch.no_grad()
    def generate_speculation(self, draft_hidden, initial_token, num_tokens: int = 5, temperature: float = 1.0,
                             top_k: int = 50, generator=None):
        """Roll ``num_tokens`` ahead from one drafter state ``[B,1,H]``; returns ``(tokens, probs)``.
        ``temperature <= 0`` is greedy."""
        toks, probs, cur = [], [], initial_token
        hist = None
        for _ in range(num_tokens):
            e = self.token_embedding(cur[:, None])
            hist = e if hist is None else torch.cat([hist, e], 1)
            logits = self(draft_hidden.expand(-1, hist.shape[1], -1), input_embeds=hist)["logits"][:, -1].float()
            if temperature <= 0:
                p = F.softmax(logits, -1)
                cur = logits.argmax(-1)
            else:
                logits = logits / temperature
                if top_k > 0:
                    kth = logits.topk(min(top_k, logits.shape[-1]))[0][:, -1, None]
                    logits = logits.masked_fill(logits < kth, float("-inf"))
                p = F.softmax(logits, -1)
                cur = torch.multinomial(p, 1, generator=generator)[:, 0]
            toks.append(cur)
            probs.append(p.gather(1, cur[:, None])[:, 0])
        return torch.stack(toks, 1), torch.stack(probs, 1)


# ------------------------------------------------------- projection / logits / distil
@dataclasses.dataclass
class TokenProjectionConfig(TokenAlignmentConfig):
    projection_type: str = "mlp"
    bottleneck_dim: int = 512
    contrastive_weight: float = 0.1


class TokenProjectionModule(BaseTokenAlignmentModule):
    """Hidden-state projection + a learned vocabulary map.  The map is stored as a
    dense ``[target_vocab, draft_vocab]`` matrix initialised to the identity and applied
    by row gather (a one-hot matmul would move 4 GB per batch at 32k x 32k)."""
    config_class = TokenProjectionConfig

    def __init__(self, config: Optional[TokenProjectionConfig] = None, **kw):
        super().__init__(config or TokenProjectionConfig(**kw))
        c = self.config
        if c.projection_type == "linear":
            self.projection = nn.Linear(c.draft_hidden_size, c.target_hidden_size)
        elif c.projection_type == "mlp":
            self.projection = MLP(c.draft_hidden_size, c.hidden_dim, c.target_hidden_size, c.num_layers, c.dropout,
                                  c.use_layer_norm)
        elif c.projection_type == "bottleneck":
            self.projection = nn.Sequential(nn.Linear(c.draft_hidden_size, c.bottleneck_dim),
                                            nn.LayerNorm(c.bottleneck_dim), nn.GELU(),
                                            nn.Linear(c.bottleneck_dim, c.target_hidden_size))
        else:
            raise ValueError(f"unknown projection type {c.projection_type}")
        self.token_mapping = nn.Linear(c.draft_vocab_size, c.target_vocab_size, bias=False)
        with torch.no_grad():
            self.token_mapping.weight.zero_()
            n = min(c.draft_vocab_size, c.target_vocab_size)
            self.token_mapping.weight[:n, :n].fill_diagonal_(1.0)

    def forward(self, draft_embeddings=None, draft_one_hot=None):
        if draft_embeddings is not None:
            return self.projection(draft_embeddings)
        if draft_one_hot is not None:
            return self.token_mapping(draft_one_hot)
        raise ValueError("pass draft_embeddings or draft_one_hot")

    def token_logits(self, draft_tokens: torch.Tensor) -> torch.Tensor:
        return F.embedding(draft_tokens, self.token_mapping.weight.t())

    def project_tokens(self, draft_tokens: torch.Tensor) -> torch.Tensor:
        return F.softmax(self.token_logits(draft_tokens) / self.config.temperature, -1)

    def compute_loss(self, draft_hidden, target_hidden, draft_tokens=None, target_tokens=None, mask=None):
        proj = self.projection(draft_hidden)
        se = ((proj - target_hidden) ** 2).mean(-1)
        cos = F.cosine_similarity(proj, target_hidden, dim=-1)
        if mask is not None:
            denom = mask.sum().clamp(min=1)
            mse, cos_mean = (se * mask).sum() / denom, (cos * mask).sum() / denom
        else:
            mse, cos_mean = se.mean(), cos.mean()
        tok = proj.new_zeros(())
        if draft_tokens is not None and target_tokens is not None:
            lg = self.token_logits(draft_tokens) / self.config.temperature
            tok = F.cross_entropy(lg.reshape(-1, lg.shape[-1]), target_tokens.reshape(-1), ignore_index=PAD_ID)
        total = mse + 0.5 * (1 - cos_mean) + self.config.contrastive_weight * tok
        return {"total_loss": total, "mse_loss": mse, "cos_loss": 1 - cos_mean, "token_loss": tok, "cos_sim": cos_mean}


@dataclasses.dataclass
class LogitAlignmentConfig(TokenAlignmentConfig):
    use_top_k_logits: bool = True
    top_k: int = 100
    kl_temperature: float = 2.0
    label_smoothing: float = 0.0


class LogitAlignmentModule(BaseTokenAlignmentModule):
    config_class = LogitAlignmentConfig

    def __init__(self, config: Optional[LogitAlignmentConfig] = None, **kw):
        super().__init__(config or LogitAlignmentConfig(**kw))
        c = self.config
        d_in = c.top_k if c.use_top_k_logits else c.draft_vocab_size
        d_out = c.top_k if c.use_top_k_logits else c.target_vocab_size
        norm = (lambda: nn.LayerNorm(c.hidden_dim)) if c.use_layer_norm else nn.Identity
        self.logit_transform = nn.Sequential(
            nn.Linear(d_in, c.hidden_dim), norm(), nn.GELU(), nn.Dropout(c.dropout),
            nn.Linear(c.hidden_dim, c.hidden_dim), norm(), nn.GELU(), nn.Dropout(c.dropout),
            nn.Linear(c.hidden_dim, d_out))
        self.residual_scale = nn.Parameter(torch.full((1,), 0.1)) if c.use_residual else None
        self.temperature = nn.Parameter(torch.full((1,), float(c.temperature)))

    def forward(self, draft_logits, draft_top_k_indices=None):
        out = self.logit_transform(draft_logits)
        if self.residual_scale is not None and out.shape == draft_logits.shape:
            out = out + self.residual_scale * draft_logits
        return out / self.temperature.clamp(min=0.1)

    def compute_loss(self, draft_logits, target_logits, mask=None):
        aligned = self(draft_logits)
        t = self.config.kl_temperature
        kl = F.kl_div(F.log_softmax(aligned / t, -1), F.softmax(target_logits / t, -1), reduction="none").sum(-1)
        kl = (kl * mask).sum() / mask.sum().clamp(min=1) if mask is not None else kl.mean()
        kl = kl * t * t
        mse = F.mse_loss(aligned, target_logits)
        return {"total_loss": kl + 0.1 * mse, "kl_loss": kl, "mse_loss": mse}


@dataclasses.dataclass
class SequenceDistillationConfig(TokenAlignmentConfig):
    num_transformer_layers: int = 2
    num_attention_heads: int = 8
    label_smoothing: float = 0.1


class SequenceDistillationModule(BaseTokenAlignmentModule):
    config_class = SequenceDistillationConfig

    def __init__(self, config: Optional[SequenceDistillationConfig] = None, **kw):
        super().__init__(config or SequenceDistillationConfig(**kw))
        c = self.config
        self.input_projection = nn.Linear(c.draft_hidden_size, c.hidden_dim)
        self.transformer = CausalEncoder(c.hidden_dim, c.num_attention_heads, c.hidden_dim * 4,
                                         c.num_transformer_layers, c.dropout)
        self.output_projection = nn.Sequential(nn.LayerNorm(c.hidden_dim),
                                               nn.Linear(c.hidden_dim, c.target_vocab_size))

    def forward(self, draft_hidden, attention_mask=None):
        pad = (attention_mask == 0) if attention_mask is not None else None
        return self.output_projection(self.transformer(self.input_projection(draft_hidden), pad))

    def compute_loss(self, draft_hidden, target_tokens, attention_mask=None):
        logits = self(draft_hidden, attention_mask)
        ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), target_tokens.reshape(-1), ignore_index=PAD_ID,
                             label_smoothing=self.config.label_smoothing)
        with torch.no_grad():
            acc, _, _ = masked_token_accuracy(logits, target_tokens)
        return {"total_loss": ce, "ce_loss": ce, "accuracy": torch.tensor(acc)}


# --------------------------------------------------------------------------- data
def _pad_ids(ids: Sequence[int], n: int) -> List[int]:
    ids = list(ids)[:n]
    return ids + [PAD_ID] * (n - len(ids))


class TokenAlignmentDataset(torch.utils.data.Dataset):
    FIELDS = ("draft_logits", "target_logits", "draft_hidden", "target_hidden")

    def __init__(self, draft_tokens, target_tokens, draft_logits=None, target_logits=None, draft_hidden=None,
                 target_hidden=None, sample_ids: Optional[List[str]] = None):
        if len(draft_tokens) != len(target_tokens):
            raise ValueError("draft/target token counts differ")
        self.draft_tokens, self.target_tokens = draft_tokens, target_tokens
        self.draft_logits, self.target_logits = draft_logits, target_logits
        self.draft_hidden, self.target_hidden = draft_hidden, target_hidden
        self.sample_ids = sample_ids or [f"sample_{i}" for i in range(len(draft_tokens))]

    def __len__(self):
        return len(self.draft_tokens)

    def __getitem__(self, i):
        item = {"draft_tokens": self.draft_tokens[i], "target_tokens": self.target_tokens[i],
                "sample_id": self.sample_ids[i]}
        for f in self.FIELDS:
            t = getattr(self, f)
            if t is not None:
                item[f] = t[i]
        return item

    @classmethod
    def from_json(cls, json_path: str, max_seq_len: int = 128, pad_token_id: int = PAD_ID, max_samples: int = -1):
        """Either a list of ``{egpt_token_ids, llava_token_ids, id}`` (benchmark output)
        or ``{results: [{egpt_tokens, vl_tokens, sample_idx}]}`` (token extractor output)."""
        with open(json_path) as f:
            data = json.load(f)
        rows = data.get("results", []) if isinstance(data, dict) else data
        if max_samples > 0:
            rows = rows[:max_samples]
        d, t, ids = [], [], []
        for r in rows:
            a = r.get("egpt_token_ids") or r.get("egpt_tokens") or []
            b = r.get("llava_token_ids") or r.get("vl_tokens") or []
            if not a or not b:
                continue
            d.append(_pad_ids(a, max_seq_len))
            t.append(_pad_ids(b, max_seq_len))
            ids.append(str(r.get("id", r.get("sample_idx", len(ids)))))
        return cls(torch.tensor(d, dtype=torch.long).view(-1, max_seq_len),
                   torch.tensor(t, dtype=torch.long).view(-1, max_seq_len), sample_ids=ids)

    @classmethod
    def from_directory(cls, dir_path: str):
        def opt(name):
            p = os.path.join(dir_path, name + ".pt")
            return torch.load(p, map_location="cpu", weights_only=True) if os.path.exists(p) else None
        ids = None
        if os.path.exists(os.path.join(dir_path, "sample_ids.json")):
            with open(os.path.join(dir_path, "sample_ids.json")) as f:
                ids = json.load(f)
        return cls(opt("draft_tokens"), opt("target_tokens"), *(opt(f) for f in cls.FIELDS), sample_ids=ids)

    def save(self, dir_path: str) -> None:
        os.makedirs(dir_path, exist_ok=True)
        for name in ("draft_tokens", "target_tokens") + self.FIELDS:
            t = getattr(self, name)
            if t is not None:
                torch.save(t, os.path.join(dir_path, name + ".pt"))
        with open(os.path.join(dir_path, "sample_ids.json"), "w") as f:
            json.dump(self.sample_ids, f)

    @classmethod
    def from_hidden_store(cls, store_dir: str, lm_head_weight: torch.Tensor, egpt_lm_head: Optional[torch.Tensor] = None,
                          max_seq_len: int = 64):
        """Build token + hidden pairs from a chunked hidden-state store by projecting
        through the LM heads (ids shifted by +1 so that 0 stays the pad id)."""
        from ..eval.acceptance import load_eval_data, project_to_tokens
        e, v, mask = load_eval_data(store_dir)
        e, v, mask = e[:, :max_seq_len], v[:, :max_seq_len], mask[:, :max_seq_len].long()
        vt, _ = project_to_tokens(v, lm_head_weight, None, 1)
        et, _ = project_to_tokens(e, egpt_lm_head if egpt_lm_head is not None else lm_head_weight, None, 1)
        return cls((et + 1) * mask, (vt + 1) * mask, draft_hidden=e, target_hidden=v)


# ------------------------------------------------------------------------ training
@torch.no_grad()
def compute_acceptance_metrics(model: TokenAdapter, dataset: TokenAlignmentDataset, device="cpu",
                               batch_size: int = 32) -> Dict:
    model.eval()
    correct = top5 = total = 0
    preds, tgts = [], []
    for i in range(0, len(dataset), batch_size):
        d = dataset.draft_tokens[i:i + batch_size].to(device)
        t = dataset.target_tokens[i:i + batch_size].to(device)
        logits = model(d[:, :-1])
        labels = t[:, 1:1 + logits.shape[1]]
        logits = logits[:, :labels.shape[1]]
        _, c, n = masked_token_accuracy(logits, labels)
        _, c5, _ = masked_token_accuracy(logits, labels, 5)
        correct, top5, total = correct + c, top5 + c5, total + n
        preds.extend(logits.argmax(-1).cpu().tolist())
        tgts.extend(labels.cpu().tolist())
    return {"acceptance_rate": correct / total if total else 0.0, "top5_rate": top5 / total if total else 0.0,
            "total_correct": correct, "total_tokens": total, "predictions": preds, "targets": tgts}


def compute_baseline_acceptance(dataset: TokenAlignmentDataset) -> Dict:
    rate, c, n = BaseTokenAlignmentModule.compute_acceptance_rate(dataset.draft_tokens, dataset.target_tokens)
    return {"acceptance_rate": rate, "total_correct": c, "total_tokens": n}


_BATCH_ARGS = {
    "TokenAdapter": ("draft_tokens", "target_tokens"),
    "EAGLEFusionModule": ("draft_hidden", "target_tokens", "target_hidden", "target_logits"),
    "TokenProjectionModule": ("draft_hidden", "target_hidden", "draft_tokens", "target_tokens"),
    "LogitAlignmentModule": ("draft_logits", "target_logits"),
    "SequenceDistillationModule": ("draft_hidden", "target_tokens"),
}


class AlignmentTrainer:
    """One trainer for every module above: AdamW(wd 0.01) + per-step cosine schedule,
    grad-clip 1.0, validation loss/accuracy each epoch, early stopping, best checkpoint."""

    def __init__(self, model: BaseTokenAlignmentModule, train_dataset, val_dataset=None, batch_size: int = 32,
                 learning_rate: float = 1e-4, device="cpu", seed: int = 0):
        self.model = model.to(device)
        self.train_ds, self.val_ds = train_dataset, val_dataset
        self.bs, self.device = batch_size, device
        self.args = _BATCH_ARGS[type(model).__name__]
        self.opt = torch.optim.AdamW(model.parameters(), lr=learning_rate, weight_decay=0.01)
        self.gen = torch.Generator().manual_seed(seed)
        self.training_stats = {"train_losses": [], "val_losses": [], "acceptance_rates": []}
        self.sched = None

    def _call(self, idx) -> Dict[str, torch.Tensor]:
        kw = []
        for name in self.args:
            t = getattr(self.train_ds if self._split == "train" else self.val_ds, name)
            kw.append(t[idx].to(self.device) if t is not None else None)
        while kw and kw[-1] is None:
            kw.pop()
        return self.model.compute_loss(*kw)

    def train_epoch(self) -> float:
        self.model.train()
        self._split = "train"
        perm = torch.randperm(len(self.train_ds), generator=self.gen)
        total, n = 0.0, 0
        for i in range(0, len(perm), self.bs):
            out = self._call(perm[i:i + self.bs])
            self.opt.zero_grad(set_to_none=True)
            out["total_loss"].backward()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
            self.opt.step()
            if self.sched is not None:
                self.sched.step()
            total, n = total + float(out["total_loss"].detach()), n + 1
        return total / max(n, 1)

    @torch.no_grad()
    def validate(self) -> Dict[str, float]:
        self.model.eval()
        self._split = "val"
        tot, acc, n = 0.0, 0.0, 0
        for i in range(0, len(self.val_ds), self.bs):
            out = self._call(torch.arange(i, min(i + self.bs, len(self.val_ds))))
            tot += float(out["total_loss"])
            acc += float(out.get("accuracy", out.get("cos_sim", torch.tensor(0.0))))
            n += 1
        return {"loss": tot / max(n, 1), "acceptance_rate": acc / max(n, 1)}

    def train(self, num_epochs: int, save_path: Optional[str] = None, early_stopping_patience: int = 5) -> Dict:
        steps = num_epochs * max(1, (len(self.train_ds) + self.bs - 1) // self.bs)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=steps)
        best, patience = float("inf"), 0
        for epoch in range(num_epochs):
            tl = self.train_epoch()
            self.training_stats["train_losses"].append(tl)
            if self.val_ds is None:
                if save_path:
                    self.save_checkpoint(save_path)
                continue
            vm = self.validate()
            self.training_stats["val_losses"].append(vm["loss"])
            self.training_stats["acceptance_rates"].append(vm["acceptance_rate"])
            if vm["loss"] < best:
                best, patience = vm["loss"], 0
                if save_path:
                    self.save_checkpoint(save_path)
            else:
                patience += 1
                if patience >= early_stopping_patience:
                    break
        return self.training_stats

    def save_checkpoint(self, path: str) -> None:
        self.model.save(path, training_stats=self.training_stats)


def train_token_adapter(train_dataset, val_dataset=None, config: Optional[TokenAdapterConfig] = None,
                        num_epochs: int = 10, batch_size: int = 32, learning_rate: float = 1e-4, device="cpu",
                        save_path: Optional[str] = None, early_stopping: int = 5) -> Tuple[TokenAdapter, Dict]:
    model = TokenAdapter(config)
    tr = AlignmentTrainer(model, train_dataset, val_dataset, batch_size, learning_rate, device)
    stats = tr.train(num_epochs, save_path, early_stopping)
    return model, stats


# -------------------------------------------------------------------- SD simulation
def compute_theoretical_speedup(acceptance_rate: float, gamma: int) -> float:
    """Expected tokens per verification ``(1 - a^(g+1)) / (1 - a)`` (drafting assumed free)."""
    if acceptance_rate >= 1.0:
        return float(gamma + 1)
    if acceptance_rate <= 0.0:
        return 1.0
    return (1 - acceptance_rate ** (gamma + 1)) / (1 - acceptance_rate)


@torch.no_grad()
def simulate_speculative_decoding(model: EAGLEFusionModule, draft_hidden: torch.Tensor, target_tokens: torch.Tensor,
                                  gamma: int = 5, device="cpu", batch_size: int = 32) -> Dict:
    """Walk each target sequence in windows of ``gamma`` drafts, accepting the longest
    matching prefix and one verifier token per step.  The module is run ONCE per batch,
    teacher-forced on the target tokens with causal attention (so a draft at position p
    sees the whole accepted prefix); the walk itself is index arithmetic on the
    resulting match matrix.  (The reference re-runs the module per window with only
    that window as context — evaluate_speculative.py:116-152.)"""
    model.eval().to(device)
    acc_tot = spec_tot = steps_tot = n_valid = 0
    seq_lens = []
    for i in range(0, draft_hidden.shape[0], batch_size):
        h = draft_hidden[i:i + batch_size].to(device)
        t = target_tokens[i:i + batch_size].to(device)
        pred = model(h[:, :-1], input_tokens=t[:, :-1])["logits"].argmax(-1)
        match = (pred == t[:, 1:]).cpu()            # match[b, p]: draft made at p equals token p+1
        lens = (t != PAD_ID).sum(1).cpu().tolist()
        for b, n in enumerate(lens):
            if n < gamma + 1:
                continue
            pos = steps = acc = spec = 0
            row = match[b].tolist()
            while pos < n - 1:
                w = min(gamma, n - pos - 1)
                k = 0
                while k < w and row[pos + k]:
                    k += 1
                acc, spec, steps, pos = acc + k, spec + w, steps + 1, pos + k + 1
            acc_tot, spec_tot, steps_tot, n_valid = acc_tot + acc, spec_tot + spec, steps_tot + steps, n_valid + 1
            seq_lens.append(n)
    if not n_valid:
        return {"error": "No valid samples"}
    rate = acc_tot / spec_tot if spec_tot else 0.0
    base = sum(seq_lens) / n_valid
    steps = steps_tot / n_valid
    return {"acceptance_rate": rate, "total_accepted": acc_tot, "total_speculated": spec_tot,
            "avg_accepted_per_step": acc_tot / steps_tot if steps_tot else 0.0,
            "theoretical_speedup": compute_theoretical_speedup(rate, gamma),
            "simulated_speedup": base / steps if steps else 1.0, "baseline_steps": base,
            "speculative_steps": steps}


def evaluate_at_different_gammas(model, draft_hidden, target_tokens, device="cpu",
                                 gammas: Sequence[int] = (1, 2, 3, 4, 5, 7, 10)) -> Dict:
    return {f"gamma_{g}": simulate_speculative_decoding(model, draft_hidden, target_tokens, g, device)
            for g in gammas}
