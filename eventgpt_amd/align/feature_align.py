"""Vision-feature alignment library: map event-encoder features into the space of an
RGB (CLIP) encoder so that a drafter's visual features can stand in for a verifier's.

Strategies (reference: feasible/feature_alignment/):

* ``ContrastiveAlignmentModule`` — CEIA-style: MLP projector + two BN projection heads,
  learnable logit scale, symmetric InfoNCE + 0.1 x normalised MSE, optional MoCo-style
  negative queue [contrastive.py:33-291].
* ``LightweightAlignmentModule`` — linear / MLP / bottleneck adapter, optional whitening,
  scaled residual [lightweight.py:40-308].
* ``TripleModalAlignmentModule`` — event / image / text projectors into one space,
  cross-modal attention, three pairwise InfoNCE terms + inference-projector MSE,
  zero-shot classification [triple_modal.py:39-475].
* ``ReconstructionBridgingModule`` — event voxel grid ``[B, bins, H, W]`` -> recurrent
  conv encoder-decoder (E2VID-shaped) -> grey image -> CLIP preprocessing -> CLIP encoder
  [reconstruction.py:36-527].  The voxel grid itself comes from
  ``eventgpt_amd.ops.event_ops.voxel_grid_gpu`` / ``data.events.voxel_grid``.
* shared base (InfoNCE, retrieval R@k, checkpoint IO), ``AlignmentTrainer``,
  ``AlignmentDataset``, feasibility test, strategy selector, diagnosis
  [base.py:15-344, utils.py:27-770].

Differences: whitening uses an eigendecomposition of the running covariance computed
once per statistics update (not an SVD per forward); reconstruction latency is
measured with device events when the module runs on a GPU; checkpoints store config
dicts, not pickled dataclasses.
"""
from __future__ import annotations

import dataclasses
import math
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclasses.dataclass
class AlignmentConfig:
    event_dim: int = 768
    target_dim: int = 1024
    hidden_dim: int = 512
    num_layers: int = 2
    dropout: float = 0.1
    temperature: float = 0.07
    use_layer_norm: bool = True
    freeze_event_encoder: bool = False
    freeze_target_encoder: bool = True

    def to_dict(self):
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, d):
        names = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names})


@dataclasses.dataclass
class AlignmentMetrics:
    feature_similarity: float
    acceptance_rate: float
    retrieval_recall_at_1: float
    retrieval_recall_at_5: float
    retrieval_recall_at_10: float
    kl_divergence: Optional[float] = None
    training_loss: Optional[float] = None


def info_nce(a: torch.Tensor, b: torch.Tensor, scale) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric InfoNCE over matched rows of two L2-normalised sets; returns (loss, logits)."""
    logits = scale * a @ b.t()
    labels = torch.arange(a.shape[0], device=a.device)
    return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2, logits


def _freeze(m: Optional[nn.Module]) -> None:
    if m is not None:
        for p in m.parameters():
            p.requires_grad = False


class MLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim=512, num_layers=2, dropout=0.1, use_layer_norm=True,
                 activation="gelu"):
        super().__init__()
        act = nn.ReLU if activation == "relu" else nn.GELU
        if num_layers <= 1:
            mods = [nn.Linear(input_dim, output_dim)]
        else:
            mods, d = [], input_dim
            for _ in range(num_layers - 1):
                mods.append(nn.Linear(d, hidden_dim))
                if use_layer_norm:
                    mods.append(nn.LayerNorm(hidden_dim))
                mods += [act(), nn.Dropout(dropout)]
                d = hidden_dim
            mods.append(nn.Linear(d, output_dim))
        self.mlp = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp(x)


class ProjectionHead(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim=512, use_batch_norm=True):
        super().__init__()
        self.projection = nn.Sequential(nn.Linear(input_dim, hidden_dim),
                                        nn.BatchNorm1d(hidden_dim) if use_batch_norm else nn.Identity(),
                                        nn.ReLU(inplace=True), nn.Linear(hidden_dim, output_dim))

    def forward(self, x):
        return self.projection(x)


class FeatureAdapter(nn.Module):
    def __init__(self, input_dim, output_dim, use_bias=True, use_scale=True):
        super().__init__()
        self.linear = nn.Linear(input_dim, output_dim, bias=use_bias)
        nn.init.xavier_uniform_(self.linear.weight)
        if use_bias:
            nn.init.zeros_(self.linear.bias)
        self.scale = nn.Parameter(torch.ones(1)) if use_scale else None

    def forward(self, x):
        out = self.linear(x)
        return out * self.scale if self.scale is not None else out


class BaseAlignmentModule(nn.Module):
    config_class = AlignmentConfig

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.training_stats = {"losses": [], "similarities": [], "epochs": 0}

    def align_features(self, event_features, target_features=None):
        return self(event_features), target_features

    @staticmethod
    def compute_similarity(a, b, normalize=True):
        if normalize:
            a, b = F.normalize(a, dim=-1), F.normalize(b, dim=-1)
        return a @ b.t()

    def compute_contrastive_loss(self, event_features, target_features, temperature: Optional[float] = None):
        t = self.config.temperature if temperature is None else temperature
        return info_nce(F.normalize(event_features, dim=-1), F.normalize(target_features, dim=-1), 1.0 / t)[0]

    @staticmethod
    def compute_mse_loss(a, b):
        return F.mse_loss(a, b)

    @staticmethod
    def evaluate_retrieval(event_features, target_features, k_values: Sequence[int] = (1, 5, 10)) -> Dict[str, float]:
        sim = F.normalize(event_features, dim=-1) @ F.normalize(target_features, dim=-1).t()
        n = sim.shape[0]
        # rank of the matched column in each row
        rank = (sim > sim.diagonal()[:, None]).sum(1)
        return {f"R@{k}": float((rank < min(k, n)).float().mean()) for k in k_values}

    def save_checkpoint(self, filepath: str, optimizer=None) -> None:
        os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
        ck = {"model_state_dict": self.state_dict(), "config": self.config.to_dict(),
              "module": type(self).__name__, "training_stats": self.training_stats}
        if optimizer is not None:
            ck["optimizer_state_dict"] = optimizer.state_dict()
        torch.save(ck, filepath)

    def load_checkpoint(self, filepath: str, optimizer=None) -> None:
        ck = torch.load(filepath, map_location="cpu", weights_only=True)
        self.load_state_dict(ck["model_state_dict"])
        self.training_stats = ck.get("training_stats", self.training_stats)
        if optimizer is not None and "optimizer_state_dict" in ck:
            optimizer.load_state_dict(ck["optimizer_state_dict"])


# ----------------------------------------------------------------- contrastive
@dataclasses.dataclass
class ContrastiveAlignmentConfig(AlignmentConfig):
    projection_dim: int = 256
    queue_size: int = 65536
    momentum: float = 0.999
    use_hard_negatives: bool = True
    hard_negative_weight: float = 0.5
    warmup_epochs: int = 5
    use_queue: bool = False


class ContrastiveAlignmentModule(BaseAlignmentModule):
    config_class = ContrastiveAlignmentConfig

    def __init__(self, config: Optional[ContrastiveAlignmentConfig] = None, event_encoder=None, target_encoder=None):
        super().__init__(config or ContrastiveAlignmentConfig())
        c = self.config
        self.event_encoder, self.target_encoder = event_encoder, target_encoder
        self.event_projector = MLP(c.event_dim, c.target_dim, c.hidden_dim, c.num_layers, c.dropout, c.use_layer_norm)
        self.event_projection_head = ProjectionHead(c.target_dim, c.projection_dim, c.hidden_dim)
        self.target_projection_head = ProjectionHead(c.target_dim, c.projection_dim, c.hidden_dim)
        if c.use_queue:
            self.register_buffer("queue", F.normalize(torch.randn(c.projection_dim, c.queue_size), dim=0))
            self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
            self.register_buffer("queue_filled", torch.zeros(1, dtype=torch.long))
        self.logit_scale = nn.Parameter(torch.tensor(math.log(1 / c.temperature)))
        if c.freeze_event_encoder:
            _freeze(event_encoder)
        if c.freeze_target_encoder:
            _freeze(target_encoder)

    def forward(self, event_features):
        return self.event_projector(event_features)

    def encode_event(self, event_stream, use_encoder: bool = True):
        if use_encoder and self.event_encoder is not None:
            with torch.set_grad_enabled(not self.config.freeze_event_encoder):
                event_stream = self.event_encoder(event_stream)
        return self(event_stream)

    def encode_target(self, rgb_image, use_encoder: bool = True):
        if use_encoder and self.target_encoder is not None:
            with torch.set_grad_enabled(not self.config.freeze_target_encoder):
                return self.target_encoder(rgb_image)
        return rgb_image

    def compute_loss(self, event_features, target_features, return_metrics: bool = False) -> Dict[str, torch.Tensor]:
        aligned = self(event_features)
        ep = F.normalize(self.event_projection_head(aligned), dim=-1)
        tp = F.normalize(self.target_projection_head(target_features), dim=-1)
        scale = self.logit_scale.exp()
        loss, logits = info_nce(ep, tp, scale)
        if self.config.use_queue and int(self.queue_filled) > 0:
            # extra negatives for the event->target direction from earlier target projections
            neg = self.queue[:, :int(self.queue_filled)].clone().detach()
            ext = torch.cat([logits, scale * ep @ neg], 1)
            labels = torch.arange(ep.shape[0], device=ep.device)
            loss = (F.cross_entropy(ext, labels) + F.cross_entropy(logits.t(), labels)) / 2
        if self.config.use_queue and self.training:
            self._dequeue_and_enqueue(tp.detach())
        mse = F.mse_loss(F.normalize(aligned, dim=-1), F.normalize(target_features, dim=-1))
        out = {"total_loss": loss + 0.1 * mse, "contrastive_loss": loss, "mse_loss": mse,
               "logit_scale": scale.detach()}
        if return_metrics:
            with torch.no_grad():
                labels = torch.arange(ep.shape[0], device=ep.device)
                out["acc_e2t"] = (logits.argmax(1) == labels).float().mean()
                out["acc_t2e"] = (logits.argmax(0) == labels).float().mean()
                out["cosine_similarity"] = F.cosine_similarity(aligned, target_features, dim=-1).mean()
        return out

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys: torch.Tensor) -> None:
        if not self.config.use_queue:
            return
        size, n, ptr = self.config.queue_size, keys.shape[0], int(self.queue_ptr)
        idx = (ptr + torch.arange(n, device=keys.device)) % size
        self.queue[:, idx] = keys.t().to(self.queue.dtype)
        self.queue_ptr[0] = (ptr + n) % size
        self.queue_filled[0] = min(size, int(self.queue_filled) + n)


# ----------------------------------------------------------------- lightweight
@dataclasses.dataclass
class LightweightAlignmentConfig(AlignmentConfig):
    use_residual: bool = True
    use_whitening: bool = False
    finetune_on_downstream: bool = True
    adapter_type: str = "mlp"


class LinearAdapter(nn.Module):
    def __init__(self, input_dim, output_dim, use_bias=True):
        super().__init__()
        self.linear = nn.Linear(input_dim, output_dim, bias=use_bias)
        with torch.no_grad():
            self.linear.weight.zero_()
            n = min(input_dim, output_dim)
            self.linear.weight[:n, :n].fill_diagonal_(1.0)
            if use_bias:
                self.linear.bias.zero_()

    def forward(self, x):
        return self.linear(x)


class BottleneckAdapter(nn.Module):
    def __init__(self, input_dim, output_dim, bottleneck_dim=64, dropout=0.1):
        super().__init__()
        self.down_proj, self.up_proj = nn.Linear(input_dim, bottleneck_dim), nn.Linear(bottleneck_dim, output_dim)
        self.dropout = nn.Dropout(dropout)
        nn.init.kaiming_normal_(self.down_proj.weight)
        nn.init.zeros_(self.up_proj.weight)

    def forward(self, x):
        return self.up_proj(self.dropout(F.gelu(self.down_proj(x))))


class WhiteningLayer(nn.Module):
    """Zero-mean, identity-covariance features (ZCA).  The whitening matrix is derived
    from the running covariance by a symmetric eigendecomposition and cached; it is
    recomputed only after the statistics changed."""

    def __init__(self, dim: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.dim, self.eps, self.momentum = dim, eps, momentum
        self.register_buffer("running_mean", torch.zeros(dim))
        self.register_buffer("running_cov", torch.eye(dim))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._cached: Optional[torch.Tensor] = None

    @staticmethod
    def _matrix(cov: torch.Tensor, eps: float) -> torch.Tensor:
        vals, vecs = torch.linalg.eigh(cov.float() + eps * torch.eye(cov.shape[0], device=cov.device))
        return (vecs * vals.clamp_min(eps).rsqrt()) @ vecs.t()

    def forward(self, x):
        if self.training:
            mean = x.mean(0)
            xc = x - mean
            cov = xc.t() @ xc / x.shape[0]
            with torch.no_grad():
                self.running_mean.lerp_(mean.detach().to(self.running_mean.dtype), self.momentum)
                self.running_cov.lerp_(cov.detach().to(self.running_cov.dtype), self.momentum)
                self.num_batches_tracked += 1
                self._cached = None
            w = self._matrix(cov.detach(), self.eps)
        else:
            xc = x - self.running_mean
            if self._cached is None or self._cached.device != x.device:
                self._cached = self._matrix(self.running_cov, self.eps)
            w = self._cached
        return xc @ w.to(x.dtype)


class LightweightAlignmentModule(BaseAlignmentModule):
    config_class = LightweightAlignmentConfig

    def __init__(self, config: Optional[LightweightAlignmentConfig] = None, event_encoder=None):
        super().__init__(config or LightweightAlignmentConfig())
        c = self.config
        self.event_encoder = event_encoder
        if c.adapter_type == "linear":
            self.adapter = LinearAdapter(c.event_dim, c.target_dim)
        elif c.adapter_type == "mlp":
            self.adapter = MLP(c.event_dim, c.target_dim, c.hidden_dim, c.num_layers, c.dropout, c.use_layer_norm)
        elif c.adapter_type == "adapter":
            self.adapter = BottleneckAdapter(c.event_dim, c.target_dim, max(c.hidden_dim // 4, 1), c.dropout)
        else:
            raise ValueError(f"unknown adapter type {c.adapter_type}")
        self.whitening = WhiteningLayer(c.event_dim) if c.use_whitening else None
        if c.use_residual:
            self.residual_proj = nn.Identity() if c.event_dim == c.target_dim else \
                nn.Linear(c.event_dim, c.target_dim, bias=False)
        else:
            self.residual_proj = None
        self.scale = nn.Parameter(torch.ones(1))
        if c.freeze_event_encoder:
            _freeze(event_encoder)

    def forward(self, event_features):
        x = self.whitening(event_features) if self.whitening is not None else event_features
        out = self.adapter(x)
        if self.residual_proj is not None:
            out = out + 0.1 * self.residual_proj(x)
        return out * self.scale

    def compute_loss(self, event_features, target_features, return_metrics: bool = False):
        a = F.normalize(self(event_features), dim=-1)
        t = F.normalize(target_features, dim=-1)
        mse = F.mse_loss(a, t)
        cos = (a * t).sum(-1).mean()
        out = {"total_loss": mse + 0.5 * (1 - cos), "mse_loss": mse, "cosine_loss": 1 - cos}
        if return_metrics:
            out["cosine_similarity"] = cos.detach()
        return out

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


# ----------------------------------------------------------------- triple modal
@dataclasses.dataclass
class TripleModalAlignmentConfig(AlignmentConfig):
    text_dim: int = 512
    projection_dim: int = 256
    use_text_alignment: bool = True
    text_loss_weight: float = 0.5
    image_loss_weight: float = 1.0
    use_cross_modal_attention: bool = True
    num_attention_heads: int = 8
    use_prompt_tuning: bool = False


class CrossModalAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int = 8, dropout: float = 0.1):
        super().__init__()
        self.num_heads, self.dropout = num_heads, dropout
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(dim, dim) for _ in range(4))

    def forward(self, query, key, value, mask=None):
        b, d = query.shape[0], query.shape[-1]

        def split(x):
            return x.view(b, -1, self.num_heads, d // self.num_heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.q_proj(query)), split(self.k_proj(key)),
                                           split(self.v_proj(value)),
                                           attn_mask=None if mask is None else mask.bool(),
                                           dropout_p=self.dropout if self.training else 0.0)
        return self.out_proj(o.transpose(1, 2).reshape(b, -1, d))


def _proj(in_dim, c) -> nn.Sequential:
    return nn.Sequential(nn.Linear(in_dim, c.hidden_dim), nn.LayerNorm(c.hidden_dim), nn.GELU(),
                         nn.Dropout(c.dropout), nn.Linear(c.hidden_dim, c.projection_dim))


class TripleModalEncoder(nn.Module):
    def __init__(self, config: TripleModalAlignmentConfig):
        super().__init__()
        c = self.config = config
        self.event_projector, self.image_projector = _proj(c.event_dim, c), _proj(c.target_dim, c)
        self.text_projector = _proj(c.text_dim, c) if c.use_text_alignment else None
        att = c.use_cross_modal_attention
        self.event_image_attention = CrossModalAttention(c.projection_dim, c.num_attention_heads, c.dropout) \
            if att else None
        self.event_text_attention = CrossModalAttention(c.projection_dim, c.num_attention_heads, c.dropout) \
            if att and c.use_text_alignment else None
        init = math.log(1 / c.temperature)
        self.logit_scale_ei = nn.Parameter(torch.tensor(init))
        self.logit_scale_et = nn.Parameter(torch.tensor(init))
        self.logit_scale_it = nn.Parameter(torch.tensor(init))

    def encode_event(self, x):
        return F.normalize(self.event_projector(x), dim=-1)

    def encode_image(self, x):
        return F.normalize(self.image_projector(x), dim=-1)

    def encode_text(self, x):
        if self.text_projector is None:
            raise ValueError("text projector not initialised")
        return F.normalize(self.text_projector(x), dim=-1)

    def forward(self, event_features, image_features=None, text_features=None) -> Dict[str, torch.Tensor]:
        enc = {"event": self.encode_event(event_features)}
        ev = enc["event"]
        if image_features is not None:
            enc["image"] = im = self.encode_image(image_features)
            if self.event_image_attention is not None:
                att = self.event_image_attention(ev[:, None], im[:, None], im[:, None])[:, 0]
                enc["event_attended"] = F.normalize(ev + 0.1 * att, dim=-1)
        if text_features is not None and self.text_projector is not None:
            enc["text"] = tx = self.encode_text(text_features)
            if self.event_text_attention is not None:
                att = self.event_text_attention(ev[:, None], tx[:, None], tx[:, None])[:, 0]
                enc["event_text_attended"] = F.normalize(ev + 0.1 * att, dim=-1)
        return enc


class TripleModalAlignmentModule(BaseAlignmentModule):
    config_class = TripleModalAlignmentConfig

    def __init__(self, config: Optional[TripleModalAlignmentConfig] = None, event_encoder=None, image_encoder=None,
                 text_encoder=None):
        super().__init__(config or TripleModalAlignmentConfig())
        c = self.config
        self.event_encoder, self.image_encoder, self.text_encoder = event_encoder, image_encoder, text_encoder
        self.triple_encoder = TripleModalEncoder(c)
        self.inference_projector = MLP(c.event_dim, c.target_dim, c.hidden_dim, c.num_layers, c.dropout)
        if c.freeze_event_encoder:
            _freeze(event_encoder)
        if c.freeze_target_encoder:
            _freeze(image_encoder)
        _freeze(text_encoder)

    def forward(self, event_features):
        return self.inference_projector(event_features)

    def encode_all_modalities(self, event_features, image_features=None, text_features=None):
        return self.triple_encoder(event_features, image_features, text_features)

    def compute_loss(self, event_features, image_features, text_features=None, return_metrics: bool = False):
        te, c = self.triple_encoder, self.config
        enc = self.encode_all_modalities(event_features, image_features, text_features)
        ev, im = enc["event"], enc["image"]
        loss_ei, logits_ei = info_nce(ev, im, te.logit_scale_ei.exp())
        out = {"loss_event_image": loss_ei}
        total = c.image_loss_weight * loss_ei
        if "text" in enc:
            tx = enc["text"]
            loss_et, logits_et = info_nce(ev, tx, te.logit_scale_et.exp())
            loss_it, _ = info_nce(im, tx, te.logit_scale_it.exp())
            out.update(loss_event_text=loss_et, loss_image_text=loss_it)
            total = total + c.text_loss_weight * loss_et + 0.5 * loss_it
        mse = F.mse_loss(F.normalize(self.inference_projector(event_features), dim=-1),
                         F.normalize(image_features, dim=-1))
        out["loss_mse"] = mse
        out["total_loss"] = total + 0.1 * mse
        if return_metrics:
            with torch.no_grad():
                labels = torch.arange(ev.shape[0], device=ev.device)
                out["acc_event_image"] = (logits_ei.argmax(1) == labels).float().mean()
                out["cosine_similarity"] = (ev * im).sum(-1).mean()
                if "text" in enc:
                    out["acc_event_text"] = (logits_et.argmax(1) == labels).float().mean()
        return out

    @torch.no_grad()
    def zero_shot_classify(self, event_features, class_text_features):
        ev = self.triple_encoder.encode_event(event_features)
        return (ev @ self.triple_encoder.encode_text(class_text_features).t()).argmax(-1)


# --------------------------------------------------- reconstruction bridging (E2VID)
@dataclasses.dataclass
class ReconstructionBridgingConfig(AlignmentConfig):
    e2vid_type: str = "e2vid"
    reconstruction_size: Tuple[int, int] = (224, 224)
    use_temporal_aggregation: bool = True
    num_temporal_bins: int = 5
    finetune_e2vid: bool = False
    cache_reconstructions: bool = False
    normalize_output: bool = True


class ConvGRUCell(nn.Module):
    """Convolutional GRU; reset and update gates come from one fused convolution."""

    def __init__(self, input_dim: int, hidden_dim: int, kernel_size: int = 3):
        super().__init__()
        self.hidden_dim = hidden_dim
        p = kernel_size // 2
        self.gates = nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, kernel_size, padding=p)
        self.candidate = nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, padding=p)

    def forward(self, x, hidden=None):
        if hidden is None:
            hidden = x.new_zeros(x.shape[0], self.hidden_dim, x.shape[2], x.shape[3])
        reset, update = torch.sigmoid(self.gates(torch.cat([x, hidden], 1))).chunk(2, 1)
        cand = torch.tanh(self.candidate(torch.cat([x, reset * hidden], 1)))
        return torch.lerp(hidden, cand, update)


class SimpleE2VIDNet(nn.Module):
    """3 conv-BN-ReLU encoder stages (stride 1,2,2) -> ConvGRU -> 2 transposed-conv decoder
    stages -> 3x3 conv -> sigmoid grey image at the input resolution (H, W multiples of 4)."""

    def __init__(self, num_input_bins: int = 5, base_channels: int = 32):
        super().__init__()
        self.num_input_bins, c = num_input_bins, base_channels

        def down(i, o, s):
            return [nn.Conv2d(i, o, 3, stride=s, padding=1), nn.BatchNorm2d(o), nn.ReLU(inplace=True)]

        def up(i, o):
            return [nn.ConvTranspose2d(i, o, 4, stride=2, padding=1), nn.BatchNorm2d(o), nn.ReLU(inplace=True)]
        self.encoder = nn.Sequential(*down(num_input_bins, c, 1), *down(c, 2 * c, 2), *down(2 * c, 4 * c, 2))
        self.recurrent = ConvGRUCell(4 * c, 4 * c, 3)
        self.decoder = nn.Sequential(*up(4 * c, 2 * c), *up(2 * c, c), nn.Conv2d(c, 1, 3, padding=1), nn.Sigmoid())

    def forward(self, event_voxel, hidden_state=None):
        h = self.recurrent(self.encoder(event_voxel), hidden_state)
        return self.decoder(h), h


class E2VIDWrapper(nn.Module):
    """Stateful wrapper: keeps the recurrent state between calls of a stream."""

    def __init__(self, e2vid_type: str = "e2vid", pretrained_path: Optional[str] = None, num_bins: int = 5,
                 model: Optional[nn.Module] = None):
        super().__init__()
        if e2vid_type not in ("e2vid", "hypere2vid", "custom"):
            raise ValueError(f"unknown E2VID type {e2vid_type}")
        self.e2vid_type = e2vid_type
        self.model = model if model is not None else (None if e2vid_type == "custom" else SimpleE2VIDNet(num_bins))
        if pretrained_path and self.model is not None:
            sd = torch.load(pretrained_path, map_location="cpu", weights_only=True)
            self.model.load_state_dict(sd.get("model_state_dict", sd) if isinstance(sd, dict) else sd, strict=False)
        self.hidden_state = None

    def forward(self, event_voxel, reset_state: bool = False):
        if self.model is None:
            raise ValueError("E2VID model not initialised")
        if reset_state or (self.hidden_state is not None and self.hidden_state.shape[0] != event_voxel.shape[0]):
            self.hidden_state = None
        out = self.model(event_voxel, self.hidden_state)
        if isinstance(out, tuple):
            out, self.hidden_state = out
            self.hidden_state = self.hidden_state.detach()
        return out

    def reset_states(self) -> None:
        self.hidden_state = None


class ReconstructionBridgingModule(BaseAlignmentModule):
    config_class = ReconstructionBridgingConfig

    def __init__(self, config: Optional[ReconstructionBridgingConfig] = None, e2vid_model=None, clip_encoder=None,
                 e2vid_path: Optional[str] = None):
        super().__init__(config or ReconstructionBridgingConfig())
        c = self.config
        self.e2vid = e2vid_model if e2vid_model is not None else \
            E2VIDWrapper(c.e2vid_type, e2vid_path, c.num_temporal_bins)
        self.clip_encoder = clip_encoder
        self.gray_to_rgb = nn.Conv2d(1, 3, 1, bias=False)
        nn.init.constant_(self.gray_to_rgb.weight, 1.0)
        self.feature_projector = MLP(c.event_dim, c.target_dim, c.hidden_dim, c.num_layers) \
            if c.event_dim != c.target_dim else None
        self.register_buffer("mean", torch.tensor(CLIP_MEAN).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(CLIP_STD).view(1, 3, 1, 1))
        if not c.finetune_e2vid:
            _freeze(self.e2vid)
        if c.freeze_target_encoder:
            _freeze(clip_encoder)
        self.reset_latency_stats()

    def _timed(self, key: str, fn):
        gpu = self.mean.is_cuda
        if gpu:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        else:
            t0 = time.perf_counter()
        out = fn()
        if gpu:
            b.record()
            self._pending.append((key, a, b))
        else:
            self.latency_stats[key].append((time.perf_counter() - t0) * 1e3)
        return out

    def forward(self, event_voxel):
        grey = self._timed("reconstruction_times", lambda: self.reconstruct_image(event_voxel))
        rgb = self.preprocess_image(grey)

        def encode():
            if self.clip_encoder is None:
                return rgb.flatten(1)
            with torch.set_grad_enabled(not self.config.freeze_target_encoder and torch.is_grad_enabled()):
                f = self.clip_encoder(rgb)
            f = getattr(f, "pooler_output", f)
            return f[0] if isinstance(f, tuple) else f
        feats = self._timed("encoding_times", encode)
        return self.feature_projector(feats) if self.feature_projector is not None else feats

    def reconstruct_image(self, event_voxel, reset_state: bool = False):
        with torch.set_grad_enabled(self.config.finetune_e2vid and torch.is_grad_enabled()):
            grey = self.e2vid(event_voxel, reset_state=reset_state) if isinstance(self.e2vid, E2VIDWrapper) \
                else self.e2vid(event_voxel)
        if isinstance(grey, tuple):
            grey = grey[0]
        size = tuple(self.config.reconstruction_size)
        if tuple(grey.shape[-2:]) != size:
            grey = F.interpolate(grey, size=size, mode="bilinear", align_corners=False)
        return grey

    def preprocess_image(self, gray_image):
        rgb = self.gray_to_rgb(gray_image)
        return (rgb - self.mean) / self.std if self.config.normalize_output else rgb

    def get_latency_stats(self) -> Dict[str, float]:
        for key, a, b in self._pending:
            b.synchronize()
            self.latency_stats[key].append(a.elapsed_time(b))
        self._pending = []
        out: Dict[str, float] = {}
        for key, name in (("reconstruction_times", "reconstruction"), ("encoding_times", "encoding")):
            v = self.latency_stats[key]
            if v:
                out[f"avg_{name}_ms"], out[f"max_{name}_ms"] = sum(v) / len(v), max(v)
        out["total_avg_ms"] = out.get("avg_reconstruction_ms", 0.0) + out.get("avg_encoding_ms", 0.0)
        return out

    def reset_latency_stats(self) -> None:
        self.latency_stats = {"reconstruction_times": [], "encoding_times": []}
        self._pending: List[tuple] = []

    def reset_e2vid_state(self) -> None:
        if hasattr(self.e2vid, "reset_states"):
            self.e2vid.reset_states()


def finetune_e2vid_on_events(net: SimpleE2VIDNet, voxels: torch.Tensor, frames: torch.Tensor, epochs: int = 5,
                             lr: float = 1e-3, batch_size: int = 8) -> List[float]:
    """Supervised fine-tune of the reconstruction net on (voxel grid, grey frame) pairs; L1 loss."""
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    losses = []
    net.train()
    for _ in range(epochs):
        tot, n = 0.0, 0
        for i in range(0, voxels.shape[0], batch_size):
            out, _ = net(voxels[i:i + batch_size])
            loss = F.l1_loss(out, frames[i:i + batch_size])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            tot, n = tot + float(loss.detach()), n + 1
        losses.append(tot / max(n, 1))
    return losses


# --------------------------------------------------------------- data + training
class AlignmentDataset(torch.utils.data.Dataset):
    """Paired pre-extracted features ``{event_features, rgb_features[, text_features]}``."""

    def __init__(self, event_features, rgb_features, text_features=None, normalize: bool = False):
        if len(event_features) != len(rgb_features):
            raise ValueError("event / rgb feature counts differ")
        if normalize:
            event_features, rgb_features = F.normalize(event_features, dim=-1), F.normalize(rgb_features, dim=-1)
        self.event_features, self.rgb_features, self.text_features = event_features, rgb_features, text_features

    def __len__(self):
        return len(self.event_features)

    def __getitem__(self, i):
        item = {"event_features": self.event_features[i], "rgb_features": self.rgb_features[i]}
        if self.text_features is not None:
            item["text_features"] = self.text_features[i]
        return item

    @classmethod
    def from_directory(cls, dir_path: str, normalize: bool = False):
        def load(name):
            p = os.path.join(dir_path, name)
            return torch.load(p, map_location="cpu", weights_only=True) if os.path.exists(p) else None
        return cls(load("event_features.pt"), load("rgb_features.pt"), load("text_features.pt"), normalize)

    def save(self, dir_path: str) -> None:
        os.makedirs(dir_path, exist_ok=True)
        torch.save(self.event_features, os.path.join(dir_path, "event_features.pt"))
        torch.save(self.rgb_features, os.path.join(dir_path, "rgb_features.pt"))
        if self.text_features is not None:
            torch.save(self.text_features, os.path.join(dir_path, "text_features.pt"))

    def split(self, train_ratio: float = 0.8, seed: int = 42):
        perm = torch.randperm(len(self), generator=torch.Generator().manual_seed(seed))
        k = int(len(self) * train_ratio)

        def sub(idx):
            return AlignmentDataset(self.event_features[idx], self.rgb_features[idx],
                                    self.text_features[idx] if self.text_features is not None else None)
        return sub(perm[:k]), sub(perm[k:])


class AlignmentTrainer:
    """AdamW + linear warm-up/cosine decay, grad-clip 1.0, per-epoch validation
    (loss, cosine, R@k), best checkpoint, early stopping — for every strategy above."""

    def __init__(self, model: BaseAlignmentModule, train_dataset: AlignmentDataset, val_dataset=None,
                 batch_size: int = 64, learning_rate: float = 1e-3, weight_decay: float = 0.01, device="cpu",
                 warmup_epochs: int = 0, seed: int = 0):
        self.model = model.to(device)
        self.train_ds, self.val_ds, self.bs, self.device = train_dataset, val_dataset, batch_size, device
        self.opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=learning_rate,
                                     weight_decay=weight_decay)
        self.base_lr, self.warmup = learning_rate, warmup_epochs
        self.gen = torch.Generator().manual_seed(seed)
        self.history: List[Dict[str, float]] = []

    def _loss(self, ds, idx, metrics=False):
        e, r = ds.event_features[idx].to(self.device), ds.rgb_features[idx].to(self.device)
        if isinstance(self.model, TripleModalAlignmentModule):
            t = ds.text_features[idx].to(self.device) if ds.text_features is not None else None
            return self.model.compute_loss(e, r, t, return_metrics=metrics)
        return self.model.compute_loss(e, r, return_metrics=metrics)

    def _set_lr(self, epoch: int, num_epochs: int) -> None:
        if epoch < self.warmup:
            lr = self.base_lr * (epoch + 1) / self.warmup
        else:
            t = (epoch - self.warmup) / max(num_epochs - self.warmup, 1)
            lr = self.base_lr * 0.5 * (1 + math.cos(math.pi * t))
        for g in self.opt.param_groups:
            g["lr"] = lr

    def train_epoch(self, epoch: int = 0) -> Dict[str, float]:
        self.model.train()
        perm = torch.randperm(len(self.train_ds), generator=self.gen)
        tot, n = 0.0, 0
        for i in range(0, len(perm), self.bs):
            idx = perm[i:i + self.bs]
            if idx.numel() < 2:          # contrastive terms and BatchNorm need at least two rows
                continue
            out = self._loss(self.train_ds, idx)
            self.opt.zero_grad(set_to_none=True)
            out["total_loss"].backward()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), 1.0)
            self.opt.step()
            tot, n = tot + float(out["total_loss"].detach()), n + 1
        return {"loss": tot / max(n, 1)}

    @torch.no_grad()
    def validate(self) -> Dict[str, float]:
        self.model.eval()
        ds = self.val_ds
        out = self._loss(ds, torch.arange(len(ds)), metrics=True)
        aligned = self.model(ds.event_features.to(self.device))
        res = {"loss": float(out["total_loss"]), "cosine_similarity": float(out.get("cosine_similarity", 0.0))}
        if aligned.shape[-1] == ds.rgb_features.shape[-1]:
            res.update(self.model.evaluate_retrieval(aligned, ds.rgb_features.to(self.device)))
        return res

    def train(self, num_epochs: int, save_path: Optional[str] = None, early_stopping_patience: int = 10):
        best, patience = float("inf"), 0
        for epoch in range(num_epochs):
            self._set_lr(epoch, num_epochs)
            rec = {"epoch": epoch + 1, "train_loss": self.train_epoch(epoch)["loss"]}
            if self.val_ds is not None:
                rec.update({"val_" + k: v for k, v in self.validate().items()})
                if rec["val_loss"] < best:
                    best, patience = rec["val_loss"], 0
                    if save_path:
                        self.model.save_checkpoint(save_path, self.opt)
                else:
                    patience += 1
            elif save_path:
                self.model.save_checkpoint(save_path, self.opt)
            self.history.append(rec)
            self.model.training_stats["losses"].append(rec["train_loss"])
            self.model.training_stats["epochs"] = epoch + 1
            if patience >= early_stopping_patience:
                break
        return self.history


_STRATEGIES = {"contrastive": (ContrastiveAlignmentModule, ContrastiveAlignmentConfig),
               "lightweight": (LightweightAlignmentModule, LightweightAlignmentConfig),
               "triple_modal": (TripleModalAlignmentModule, TripleModalAlignmentConfig)}


def create_alignment(strategy: str, **config_kw) -> BaseAlignmentModule:
    cls, cfg = _STRATEGIES[strategy]
    return cls(cfg(**config_kw))


def train_alignment(strategy: str, train_dataset, val_dataset=None, num_epochs: int = 20, batch_size: int = 64,
                    learning_rate: float = 1e-3, device="cpu", save_path: Optional[str] = None, **config_kw):
    """One entry point for the per-strategy ``train_*_alignment`` helpers of the reference."""
    model = create_alignment(strategy, **config_kw)
    tr = AlignmentTrainer(model, train_dataset, val_dataset, batch_size, learning_rate, device=device)
    return model, tr.train(num_epochs, save_path)


# ------------------------------------------------------------------- utilities
def _pooled(f):
    f = getattr(f, "pooler_output", f)
    if isinstance(f, tuple):
        f = f[0]
    return f[:, 0] if f.dim() == 3 else f


@torch.no_grad()
def quick_feasibility_test(event_encoder, clip_encoder, test_data, device="cpu", verbose: bool = False) -> Dict:
    """Initial similarity of the two feature spaces: <0.3 strong alignment needed, <0.6 medium, else light."""
    sims, l2 = [], []
    for batch in test_data:
        ev = batch.get("event_stream", batch.get("event_features"))
        rgb = batch.get("rgb_image", batch.get("rgb_features"))
        if ev is None or rgb is None:
            continue
        a = F.normalize(_pooled(event_encoder(ev.to(device))).float(), dim=-1)
        b = F.normalize(_pooled(clip_encoder(rgb.to(device))).float(), dim=-1)
        if a.shape[-1] != b.shape[-1]:
            d = min(a.shape[-1], b.shape[-1])
            a, b = F.normalize(a[..., :d], dim=-1), F.normalize(b[..., :d], dim=-1)
        sims += (a * b).sum(-1).tolist()
        l2 += (a - b).norm(dim=-1).tolist()
    t = torch.tensor(sims) if sims else torch.zeros(1)
    avg = float(t.mean())
    level = "strong" if avg < 0.3 else ("medium" if avg < 0.6 else "light")
    rec = {"strong": "strong alignment needed (contrastive or triple-modal)",
           "medium": "medium alignment needed (lightweight fine-tune)",
           "light": "probably only a light adjustment needed"}[level]
    res = {"avg_similarity": avg, "std_similarity": float(t.std()) if t.numel() > 1 else 0.0,
           "avg_l2_distance": sum(l2) / len(l2) if l2 else 0.0, "num_samples": len(sims),
           "recommendation": rec, "needs_alignment": level}
    if verbose:
        print(f"[feasibility] cos {avg:.4f} over {len(sims)} samples -> {rec}")
    return res


@dataclasses.dataclass
class DiagnosisResult:
    feature_similarity: float
    acceptance_rate: float
    position_decay: str
    overall_quality: str
    recommendations: List[str]


@torch.no_grad()
def diagnose_alignment_quality(aligned_features: torch.Tensor, target_features: torch.Tensor,
                               acceptance_rate: Optional[float] = None) -> DiagnosisResult:
    """Grade an alignment from paired features (and a measured acceptance rate if there is
    one; otherwise the estimate ``min(0.8, 1.1 x similarity)`` is used)."""
    sim = float(F.cosine_similarity(aligned_features.float(), target_features.float(), dim=-1).mean())
    acc = min(0.8, sim * 1.1) if acceptance_rate is None else acceptance_rate
    decay = "slow" if sim > 0.7 else ("medium" if sim > 0.5 else "fast")
    if sim > 0.7 and acc > 0.65:
        quality, recs = "excellent", []
    elif sim > 0.5 and acc > 0.50:
        quality, recs = "medium", ["add training data", "try a stronger alignment method", "lower gamma"]
    else:
        quality, recs = "poor", ["use contrastive alignment", "collect more paired data",
                                 "check the event encoder"]
    return DiagnosisResult(sim, acc, decay, quality, recs)


@dataclasses.dataclass
class StrategyRecommendation:
    strategy: str
    confidence: str
    reasons: List[str]
    estimated_improvement: str
    caveats: Optional[List[str]] = None


def auto_select_alignment_strategy(has_paired_data: bool, data_size: int, has_e2vid: bool,
                                   compute_budget: str = "medium", target_performance: str = "good"):
    if target_performance == "best" and has_paired_data and data_size > 10000:
        return StrategyRecommendation("Triple-Modal Alignment", "high",
                                      ["plenty of data (>10K)", "best quality wanted", "can use image-text data"],
                                      "+30-40%", ["long training", "more GPU memory"])
    if has_paired_data and data_size > 5000:
        return StrategyRecommendation("Contrastive Alignment (CEIA)", "high",
                                      [f"enough paired data ({data_size})", "strong results", "stable training"],
                                      "+20-30%", ["1-2 days of training"])
    if has_e2vid:
        return StrategyRecommendation("Reconstruction Bridging", "medium",
                                      ["pretrained E2VID available", "no paired data needed", "quick to deploy"],
                                      "+10-20%", ["adds reconstruction latency"])
    if has_paired_data and data_size > 1000:
        return StrategyRecommendation("Lightweight Alignment + Fine-tune", "medium",
                                      [f"some paired data ({data_size})", "fast prototype"], "+10-15%",
                                      ["limited headroom"])
    return StrategyRecommendation("Lightweight Alignment", "low", ["little data", "few resources"], "+5-15%",
                                  ["validate feasibility first"])


def print_strategy_recommendation(r: StrategyRecommendation) -> None:
    print(f"strategy: {r.strategy} (confidence {r.confidence}, expected {r.estimated_improvement})")
    for x in r.reasons:
        print("  -", x)
    for x in r.caveats or []:
        print("  !", x)


@torch.no_grad()
def evaluate_alignment_comprehensive(model: BaseAlignmentModule, dataset: AlignmentDataset, device="cpu") -> Dict:
    model.eval().to(device)
    aligned = model(dataset.event_features.to(device))
    tgt = dataset.rgb_features.to(device)
    out = {"num_samples": len(dataset)}
    if aligned.shape[-1] == tgt.shape[-1]:
        cos = F.cosine_similarity(aligned, tgt, dim=-1)
        out.update(cosine_mean=float(cos.mean()), cosine_std=float(cos.std()) if cos.numel() > 1 else 0.0,
                   mse=float(F.mse_loss(F.normalize(aligned, dim=-1), F.normalize(tgt, dim=-1))),
                   **model.evaluate_retrieval(aligned, tgt))
        d = diagnose_alignment_quality(aligned, tgt)
        out.update(quality=d.overall_quality, estimated_acceptance=d.acceptance_rate)
    return out


def compare_alignment_strategies(train_ds: AlignmentDataset, val_ds: AlignmentDataset,
                                 strategies: Sequence[str] = ("lightweight", "contrastive"), num_epochs: int = 10,
                                 device="cpu", **config_kw) -> Dict[str, Dict]:
    res = {}
    for s in strategies:
        if s == "triple_modal" and train_ds.text_features is None:
            continue
        model, _ = train_alignment(s, train_ds, val_ds, num_epochs, device=device, **config_kw)
        res[s] = evaluate_alignment_comprehensive(model, val_ds, device)
        res[s]["parameters"] = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return res


@torch.no_grad()
def extract_and_save_features(event_encoder, clip_encoder, loader, out_dir: str, device="cpu") -> AlignmentDataset:
    ev, rgb = [], []
    for batch in loader:
        ev.append(_pooled(event_encoder(batch["event_stream"].to(device))).float().cpu())
        rgb.append(_pooled(clip_encoder(batch["rgb_image"].to(device))).float().cpu())
    ds = AlignmentDataset(torch.cat(ev), torch.cat(rgb))
    ds.save(out_dir)
    return ds
