"""EventChatModel — the event-stream multimodal LLM.

API surface kept from the reference so user code can switch over
(reference: model/EventChatModel.py:15-469):

  ``EventChatConfig``, ``VisualTower`` (``visual_tower_name``,
  ``event_processor``, ``visual_tower``), ``EventChatLlamaModel``
  (``visual_tower``, ``visual_projector``, ``feature_adaptor``,
  ``initialize_vision_modules``), ``EventChatModel`` (``get_model``,
  ``get_visual_tower``, ``visval_encode``, ``initialize_vision_tokenizer``,
  ``generate``, ``prepare_inputs_labels_for_multimodal`` returning the 6-tuple,
  ``model.model(...)`` / ``lm_head`` / ``resize_token_embeddings`` / ``device``),
  and ``get_spatio_temporal_features``.

Differences by design:
  * all frames of all samples go through the vision tower as ONE batch;
  * the splice is ONE gather: the host lays out, from the prompt ids, which embedding-table
    row or event-feature row every output position takes (``splice_plan``; the only device
    read is the copy of the ids themselves when they live on the GPU), and one kernel
    (``splice_rows``) fills the padded ``[B, L, H]`` batch -- no per-sample ``cat``,
    ``nonzero`` or ``int(sum)`` round trips;
  * ``event_features`` takes ``already_adapted`` — the reference applies
    ``feature_adaptor`` to cached features even when the caller already did
    (model/EventChatModel.py:346-353); the default reproduces that, the flag
    fixes it;
  * on a GPU, ``generate`` runs on ``runtime.engine.GenerationEngine`` (static
    KV cache, fused HIP kernels, captured decode graph).
"""
from __future__ import annotations

import dataclasses
import os
from typing import List, Optional, Sequence, Tuple, Union

import torch
from torch import nn

from .. import ops
from ..constants import (DEFAULT_EV_END_TOKEN, DEFAULT_EV_START_TOKEN,
                         DEFAULT_EVENT_PATCH_TOKEN, EVENT_TOKEN_INDEX, IGNORE_INDEX,
                         MAX_MULTIMODAL_LENGTH)
from ..data.clip_preprocess import EventImageProcessor
from ..utils import checkpoint as ckpt
from .clip_vit import ClipVisionModel
from .config import ClipVisionConfig, EventChatConfig
from .llama import CausalLMOutput, DecoderOutput, Linear, LlamaModel

_DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}


def get_spatio_temporal_features(features, num_temporal_tokens: Optional[int] = None):
    """``[T, S, C]`` (tensor or list of ``[S, C]``) → ``[T + S, C]``."""
    if isinstance(features, (list, tuple)):
        features = torch.stack(list(features))
    return ops.spatio_temporal_pool(features, num_temporal_tokens)


@dataclasses.dataclass
class GenerateOutput:
    sequences: torch.Tensor
    hidden_states: Optional[tuple] = None
    scores: Optional[tuple] = None


class GELU(nn.Module):
    """Marker module: the projector's GELU is fused into the preceding GEMM."""

    def forward(self, x):
        return torch.nn.functional.gelu(x)


class MLPProjector(nn.Sequential):
    """``Linear → GELU → Linear``; state-dict keys ``0.*`` and ``2.*``."""

    def __init__(self, in_dim: int, out_dim: int, depth: int = 2):
        mods: List[nn.Module] = [Linear(in_dim, out_dim)]
        for _ in range(1, depth):
            mods += [GELU(), Linear(out_dim, out_dim)]
        super().__init__(*mods)

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], GELU)
            x = mods[i](x, act="gelu" if fuse else None)
            i += 2 if fuse else 1
        return x


class VisualTower(nn.Module):
    """Frozen CLIP tower + its image processor.

    Weights are looked up under ``./checkpoints/<name>`` or ``<name>`` as a
    directory; if neither exists the tower is random-initialised with the
    geometry in ``vision_config`` (synthetic runs).
    """

    def __init__(self, visual_tower: str, vision_config: Optional[ClipVisionConfig] = None,
                 dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.visual_tower_name = visual_tower
        path = None
        for cand in (os.path.join("./checkpoints", str(visual_tower)), str(visual_tower)):
            if os.path.isdir(cand) and os.path.exists(os.path.join(cand, "config.json")):
                path = cand
                break
        if path is not None and vision_config is None:
            vision_config = ClipVisionConfig.from_pretrained(path)
        self.visual_tower = ClipVisionModel(vision_config).to(dtype)
        self.event_processor = EventImageProcessor(self.visual_tower.config.image_size)
        if path is not None:
            ckpt.load_into(self.visual_tower, path, strict=False)
        self.visual_tower.requires_grad_(False)
        self.is_loaded = path is not None


class EventChatLlamaModel(LlamaModel):
    config_class = EventChatConfig

    def __init__(self, config: EventChatConfig):
        super().__init__(config)
        vc = getattr(config, "vision_config", None)
        self.text_hidden_size = vc.hidden_size if vc is not None else 1024
        self.mlp_depth = 2
        if hasattr(config, "mm_visual_tower"):
            self.visual_tower = self.build_visual_tower(config.mm_visual_tower)
            self.visual_projector = self.build_mlp_projector(self.text_hidden_size,
                                                             config.hidden_size)
        if hasattr(config, "event_feature_adaptor"):
            self.feature_adaptor = Linear(config.hidden_size, config.hidden_size)
        if getattr(config, "use_event_qformer", False):
            raise NotImplementedError(
                "use_event_qformer: the reference never defined this module "
                "(build_event_qformer is missing upstream); no checkpoint uses it")

    def build_visual_tower(self, name) -> VisualTower:
        return VisualTower(name, getattr(self.config, "vision_config", None),
                           _DTYPES.get(str(getattr(self.config, "torch_dtype", "bfloat16")),
                                       torch.bfloat16))

    def build_mlp_projector(self, in_dim: int, out_dim: int) -> MLPProjector:
        return MLPProjector(in_dim, out_dim, self.mlp_depth)

    def initialize_vision_modules(self, model_args, fsdp=None) -> None:
        """Training-time builder; loads partial pretrained weights whose keys carry
        the ``model.feature_adaptor.`` / ``model.visual_projector.`` prefixes."""
        self.config.mm_visual_tower = model_args.vision_tower
        self.config.event_feature_adaptor = True
        self.visual_tower = self.build_visual_tower(model_args.vision_tower)
        self.text_hidden_size = self.visual_tower.visual_tower.config.hidden_size
        self.visual_projector = self.build_mlp_projector(self.text_hidden_size,
                                                         self.config.hidden_size)
        if getattr(model_args, "use_feature_adaptor", True):
            self.feature_adaptor = Linear(self.config.hidden_size, self.config.hidden_size)
        if getattr(model_args, "use_event_qformer", False):
            raise NotImplementedError("use_event_qformer is not defined upstream")
        for attr, module, prefix in (
                ("pretrain_feature_adaptor", "feature_adaptor", "model.feature_adaptor."),
                ("pretrain_mm_mlp_adapter", "visual_projector", "model.visual_projector.")):
            src = getattr(model_args, attr, None)
            if src is None:
                continue
            sd = torch.load(src, map_location="cpu", weights_only=True)
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
            getattr(self, module).load_state_dict(sd, strict=True)


_PAD = -(1 << 30)     # splice map value of a padding position


def splice_plan(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], feat_rows: Sequence[int]):
    """Layout of the multimodal splice, computed on the host from CPU tensors.

    ``input_ids [B, L]`` hold ``EVENT_TOKEN_INDEX`` where an event goes; the j-th placeholder
    (counted over the batch in order; a sample WITHOUT a placeholder also consumes one entry,
    as in the reference, model/EventChatModel.py:371-413) is replaced by ``feat_rows[j]`` rows.
    Returns ``(src [B, Lmax] int64, label_src [B, Lmax] int64, lengths)``: ``src >= 0`` is a
    token id to embed, ``-1 - r`` is row ``r`` of the concatenated event features, ``_PAD``
    is right padding; ``label_src`` is the flat index into ``labels`` (``-1`` = IGNORE_INDEX).
    Rows are truncated to ``MAX_MULTIMODAL_LENGTH``."""
    ids = input_ids.long()
    bsz, length = ids.shape
    keep = torch.ones_like(ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
    starts = [0]
    for r in feat_rows:
        starts.append(starts[-1] + int(r))
    rows, labs = [], []
    ev_i = 0
    for b in range(bsz):
        sel = torch.nonzero(keep[b]).flatten()
        row_ids = ids[b][sel]
        flat_idx = sel + b * length
        is_ev = row_ids == EVENT_TOKEN_INDEX
        if not bool(is_ev.any()):
            rows.append(row_ids)
            labs.append(flat_idx)
            ev_i += 1
            continue
        pieces, lpieces, prev = [], [], 0
        for cut in torch.nonzero(is_ev).flatten().tolist():
            pieces.append(row_ids[prev:cut])
            lpieces.append(flat_idx[prev:cut])
            if ev_i >= len(feat_rows):
                raise ValueError("more <event> placeholders than event features")
            n = starts[ev_i + 1] - starts[ev_i]
            pieces.append(-1 - torch.arange(starts[ev_i], starts[ev_i] + n))
            lpieces.append(torch.full((n,), -1, dtype=torch.long))
            ev_i += 1
            prev = cut + 1
        pieces.append(row_ids[prev:])
        lpieces.append(flat_idx[prev:])
        rows.append(torch.cat(pieces)[:MAX_MULTIMODAL_LENGTH])
        labs.append(torch.cat(lpieces)[:MAX_MULTIMODAL_LENGTH])
    lengths = [int(r.shape[0]) for r in rows]
    max_len = max(lengths)
    src = torch.full((bsz, max_len), _PAD, dtype=torch.long)
    lab = torch.full((bsz, max_len), -1, dtype=torch.long)
    for b, (r, l) in enumerate(zip(rows, labs)):
        src[b, :r.shape[0]] = r
        lab[b, :l.shape[0]] = l
    return src, lab, lengths


class EventChatModel(nn.Module):
    config_class = EventChatConfig

    def __init__(self, config: EventChatConfig):
        super().__init__()
        self.config = config
        self.model = EventChatLlamaModel(config)
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False)
        self._engine = None
        self.generation_config = None

    # ------------------------------------------------------------ construction
    @classmethod
    def from_config(cls, config: EventChatConfig, dtype=None, device=None,
                    seed: Optional[int] = 0) -> "EventChatModel":
        """Random-initialised model (N(0, 0.02) weights), built directly on ``device``."""
        dtype = dtype or _DTYPES.get(str(config.torch_dtype), torch.bfloat16)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with torch.device(device or "cpu"):
                model = cls(config)
        finally:
            torch.set_default_dtype(prev)
        model.to(dtype)
        if seed is not None:
            gen = torch.Generator(device=model.device).manual_seed(seed)
            with torch.no_grad():
                for name, p in model.named_parameters():
                    if p.ndim >= 2 or "class_embedding" in name:
                        p.normal_(0.0, 0.02, generator=gen)
                    elif name.endswith("bias"):
                        p.zero_()
                    else:
                        p.fill_(1.0)
        return model.eval()

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, config: Optional[EventChatConfig] = None,
                        device=None, strict: bool = False) -> "EventChatModel":
        config = config or EventChatConfig.from_pretrained(path)
        model = cls.from_config(config, torch_dtype, device, seed=None)
        missing, unexpected = ckpt.load_into(model, path, strict=False)
        missing = [k for k in missing if not k.startswith("model.visual_tower.")
                   or not model.get_visual_tower().is_loaded]
        if strict and (missing or unexpected):
            raise KeyError(f"missing={missing[:5]} unexpected={unexpected[:5]}")
        model.load_report = {"missing": missing, "unexpected": unexpected}
        return model.eval()

    def save_pretrained(self, path: str, tokenizer=None, max_shard_bytes: int = 5 << 30) -> None:
        self.config.save_pretrained(path)
        ckpt.save_sharded(self.state_dict(), path, max_shard_bytes)
        if tokenizer is not None and hasattr(tokenizer, "save_pretrained"):
            tokenizer.save_pretrained(path)

    # -------------------------------------------------------------- accessors
    def get_model(self) -> EventChatLlamaModel:
        return self.model

    def get_visual_tower(self) -> VisualTower:
        return self.get_model().visual_tower

    def get_input_embeddings(self) -> nn.Embedding:
        return self.model.embed_tokens

    def get_output_embeddings(self) -> nn.Linear:
        return self.lm_head

    @property
    def device(self) -> torch.device:
        return self.lm_head.weight.device

    @property
    def dtype(self) -> torch.dtype:
        return self.lm_head.weight.dtype

    def resize_token_embeddings(self, new_size: int) -> nn.Embedding:
        old = self.model.embed_tokens.weight.shape[0]
        if new_size == old:
            return self.model.embed_tokens
        keep = min(old, new_size)
        for owner, attr, is_embed in ((self.model, "embed_tokens", True), (self, "lm_head", False)):
            w_old = getattr(owner, attr).weight.data
            w_new = torch.empty(new_size, w_old.shape[1], dtype=w_old.dtype, device=w_old.device)
            w_new.normal_(0.0, 0.02)
            w_new[:keep] = w_old[:keep]
            mod = nn.Embedding(new_size, w_old.shape[1], device="meta") if is_embed else \
                Linear(w_old.shape[1], new_size, bias=False, device="meta")
            mod.weight = nn.Parameter(w_new, requires_grad=getattr(owner, attr).weight.requires_grad)
            setattr(owner, attr, mod)
        self.config.vocab_size = new_size
        self._engine = None
        return self.model.embed_tokens

    # ------------------------------------------------------------------ vision
    def visval_encode(self, event_tensor: torch.Tensor) -> torch.Tensor:
        """``[B,3,H,W]`` → projector output ``[B, 577, hidden]`` (tower is frozen)."""
        if (torch.is_grad_enabled() and event_tensor.is_cuda and self.dtype == torch.bfloat16
                and getattr(self, "train_tower_on_engine", True) and self._vision_engine_fits()):
            # a training step: the frozen tower is one graph replay of the serving engine's launch
            # sequence instead of ~300 eager launches; projector and adaptor stay on autograd
            feats = self.vision_engine().encode_tower(event_tensor)
        else:
            with torch.no_grad():
                feats = self.get_model().visual_tower.visual_tower(event_tensor).last_hidden_state
        feats = feats.detach()
        if torch.is_grad_enabled():
            feats.requires_grad_(True)
        return self.get_model().visual_projector(feats)

    def encode_event_frames(self, frames: Sequence[torch.Tensor]) -> torch.Tensor:
        """All frames of one sample in one tower batch → pooled ``[T+S, hidden]``."""
        batch = torch.stack([f.to(self.device) for f in frames])
        if (batch.is_cuda and self.dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and self._vision_engine_fits()):
            return self.vision_engine().encode(batch)
        feats = self.visval_encode(batch)
        if hasattr(self.get_model(), "feature_adaptor"):
            feats = self.get_model().feature_adaptor(feats)
        return get_spatio_temporal_features(feats)

    def initialize_vision_tokenizer(self, model_args, tokenizer) -> None:
        if model_args.mm_use_im_patch_token:
            tokenizer.add_tokens([DEFAULT_EVENT_PATCH_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
        if model_args.mm_use_im_start_end:
            n_new = tokenizer.add_tokens([DEFAULT_EV_START_TOKEN, DEFAULT_EV_END_TOKEN],
                                         special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            if n_new > 0:
                for w in (self.get_input_embeddings().weight.data,
                          self.get_output_embeddings().weight.data):
                    w[-n_new:] = w[:-n_new].float().mean(dim=0, keepdim=True).to(w.dtype)
            if getattr(model_args, "tune_mm_mlp_adapter", False):
                self.get_input_embeddings().weight.requires_grad_(True)
                self.get_output_embeddings().weight.requires_grad_(False)
            src = getattr(model_args, "pretrain_mm_mlp_adapter", None)
            if src:
                rows = torch.load(src, map_location="cpu",
                                  weights_only=True)["model.embed_tokens.weight"]
                emb = self.get_input_embeddings().weight.data
                if rows.shape == emb.shape:
                    emb[-n_new:] = rows[-n_new:]
                elif rows.shape[0] == n_new:
                    emb[-n_new:] = rows
                else:
                    raise ValueError(f"unexpected embed_tokens rows {tuple(rows.shape)} "
                                     f"for {n_new} new tokens (model {tuple(emb.shape)})")
        elif model_args.mm_use_im_patch_token and getattr(model_args, "tune_mm_mlp_adapter", False):
            self.get_input_embeddings().weight.requires_grad_(False)
            self.get_output_embeddings().weight.requires_grad_(False)

    # -------------------------------------------------------- multimodal splice
    def _event_features_from_tensors(self, event_tensors):
        if isinstance(event_tensors, (list, tuple)):
            samples = event_tensors if all(isinstance(s, (list, tuple)) for s in event_tensors) \
                else [event_tensors]
            t = len(samples[0])
            if len(samples) > 1 and torch.is_grad_enabled() and all(len(s) == t for s in samples):
                # a training batch: every sample's frames through the tower and the projector together
                # (B x T x 577 rows in one set of GEMMs), pooled per sample
                batch = torch.stack([f.to(self.device) for s in samples for f in s])
                feats = self.visval_encode(batch)
                if hasattr(self.get_model(), "feature_adaptor"):
                    feats = self.get_model().feature_adaptor(feats)
                feats = feats.view(len(samples), t, *feats.shape[1:])
                return torch.stack([get_spatio_temporal_features(feats[i]) for i in range(len(samples))])
            return torch.stack([self.encode_event_frames(s) for s in samples])
        # a plain tensor batch: projector output only, no adaptor / pooling
        return self.visval_encode(event_tensors)

    @staticmethod
    def _feature_positions(src, feat_cat, embed):
        """Flat output position of every event-feature row (int32, host) when the splice can run as the
        one-launch autograd node -- bf16 on a GPU, a frozen token table, features with a gradient, every
        feature row used exactly once -- else None."""
        if not (feat_cat.is_cuda and feat_cat.dtype == torch.bfloat16 and feat_cat.requires_grad
                and not embed.weight.requires_grad and feat_cat.shape[1] % 8 == 0):
            return None
        flat = src.reshape(-1)
        pos = torch.nonzero((flat < 0) & (flat > _PAD)).squeeze(1)
        ids = -1 - flat[pos]
        n = feat_cat.shape[0]
        if pos.numel() != n or ids.numel() == 0 or int(ids.max()) >= n or torch.unique(ids).numel() != n:
            return None
        inv = torch.empty(n, dtype=torch.int32)
        inv[ids] = pos.to(torch.int32)
        return inv

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask,
                                             past_key_values, labels, event_tensors=None,
                                             event_image_sizes=None, event_features=None,
                                             already_adapted: bool = False):
        if event_tensors is None and event_features is None:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels

        if event_features is None:
            event_features = self._event_features_from_tensors(event_tensors)
        elif not already_adapted and hasattr(self.get_model(), "feature_adaptor"):
            fa = self.get_model().feature_adaptor
            event_features = [fa(f) for f in event_features] \
                if isinstance(event_features, (list, tuple)) else fa(event_features)

        had_labels, had_mask, had_pos = labels is not None, attention_mask is not None, \
            position_ids is not None
        feats = list(event_features) if isinstance(event_features, (list, tuple)) else \
            [event_features[i] for i in range(event_features.shape[0])]
        embed = self.get_model().embed_tokens
        dev = embed.weight.device
        # layout on the host: ids are a few dozen integers (one copy if they live on the device)
        ids_h = input_ids.detach().cpu()
        mask_h = None if attention_mask is None else attention_mask.detach().bool().cpu()
        src, lab_src, lengths = splice_plan(ids_h, mask_h, [int(f.shape[0]) for f in feats])
        bsz, max_len = src.shape
        used = int((src < 0).logical_and(src > _PAD).sum()) > 0
        feat_cat = torch.cat([f.to(device=dev, dtype=embed.weight.dtype) for f in feats]) if used else \
            embed.weight.new_zeros(1, embed.weight.shape[1])
        src_d = src.to(dev)
        needs_grad = torch.is_grad_enabled() and (embed.weight.requires_grad or feat_cat.requires_grad)
        if dev.type == "cuda" and embed.weight.dtype == torch.bfloat16 and not needs_grad:
            from ..ops import hip
            embeds = hip.raw.splice_rows(embed.weight.detach(), feat_cat.detach().contiguous(),
                                         src_d.to(torch.int32).contiguous())
        elif (inv_h := self._feature_positions(src, feat_cat, embed)) is not None:
            # frozen table, features that train, every feature row placed once: one launch each way
            from ..ops import differentiable as diff
            embeds = diff.splice_rows(embed.weight.detach(), feat_cat, src_d.to(torch.int32).contiguous(),
                                      inv_h.to(dev))
        else:
            tok = embed(src_d.clamp(min=0))
            ev = feat_cat[(-1 - src_d).clamp(min=0, max=feat_cat.shape[0] - 1)]
            embeds = torch.where((src_d >= 0)[..., None], tok, ev)
            embeds = embeds * (src_d > _PAD)[..., None].to(embeds.dtype)
        # labels / mask / positions follow the same layout (host arithmetic, one upload each)
        out_mask = (src > _PAD)
        out_labels = None
        if had_labels:
            lab_h = labels.detach().cpu()
            flat = torch.cat([lab_h.reshape(-1), lab_h.new_full((1,), IGNORE_INDEX)])
            host_labels = flat[lab_src.clamp(min=-1)].reshape(bsz, max_len)
            out_labels = host_labels.to(labels.device)
            out_labels.host_copy = host_labels      # the loss selects the supervised rows on the host
        out_pos = (torch.arange(max_len)[None].expand(bsz, max_len) * out_mask).to(input_ids.device)
        return (None,
                out_pos if had_pos else None,
                out_mask.to(device=input_ids.device, dtype=attention_mask.dtype) if had_mask else None,
                past_key_values, embeds, out_labels)

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids=None, attention_mask=None, position_ids=None,
                past_key_values=None, inputs_embeds=None, labels=None, use_cache=False,
                output_hidden_states=False, event_tensors=None, event_image_sizes=None,
                event_features=None, already_adapted=False, return_dict=True,
                loss_reduction: str = "token_mean"):
        """``loss_reduction``: ``"token_mean"`` (mean over every supervised token of the batch, the HF
        convention) or ``"sequence_mean"`` (mean over sequences of each sequence's token mean: a batch
        of B samples then has the loss -- and gradients -- of B single-sample steps averaged)."""
        spliced = False
        if inputs_embeds is None and (event_tensors is not None or event_features is not None):
            spliced = True
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds,
             labels) = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, past_key_values, labels,
                event_tensors, event_image_sizes, event_features, already_adapted)
        out = self.model(input_ids=input_ids if inputs_embeds is None else None,
                         attention_mask=attention_mask, position_ids=position_ids,
                         past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                         use_cache=use_cache, output_hidden_states=output_hidden_states,
                         # the splice above right-pads (tokens masked out on entry are dropped first)
                         right_padded=spliced and labels is not None and not use_cache)
        h = out.last_hidden_state
        from ..ops import differentiable as diff
        if labels is not None and h.requires_grad and diff.linear_cross_entropy_ok(h, self.lm_head.weight):
            # training step on the GPU: the LM head runs on the supervised rows only and the loss never
            # materialises fp32 logits (``ops.differentiable.linear_cross_entropy``); ``logits`` is None
            host = getattr(labels, "host_copy", None)
            loss = diff.linear_cross_entropy(h[:, :-1], self.lm_head.weight, labels[:, 1:], IGNORE_INDEX,
                                             None if host is None else host[:, 1:],
                                             per_sequence_mean=loss_reduction == "sequence_mean")
            return CausalLMOutput(None, out.past_key_values, out.hidden_states, loss)
        logits = self.lm_head(h)
        loss = None
        if labels is not None:
            shift_logits = logits[:, :-1].float().reshape(-1, logits.shape[-1])
            shift_labels = labels[:, 1:].reshape(-1)
            if loss_reduction == "sequence_mean":
                tok = torch.nn.functional.cross_entropy(shift_logits, shift_labels, ignore_index=IGNORE_INDEX,
                                                        reduction="none").view(labels.shape[0], -1)
                n = (labels[:, 1:] != IGNORE_INDEX).sum(1)
                loss = (tok.sum(1) / n.clamp(min=1))[n > 0].mean()
            else:
                loss = torch.nn.functional.cross_entropy(shift_logits, shift_labels,
                                                         ignore_index=IGNORE_INDEX)
        return CausalLMOutput(logits, out.past_key_values, out.hidden_states, loss)

    # ----------------------------------------------------------------- generate
    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, event_tensors=None,
                 event_image_sizes=None, event_features=None, event_feature=None,
                 already_adapted: bool = False, **kwargs) -> Union[GenerateOutput, torch.Tensor]:
        """Generate from a multimodal prompt; returns ONLY the new token ids
        (the prompt is given as embeddings), ``[B, n_new]``."""
        position_ids = kwargs.pop("position_ids", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        if event_tensors is None and event_features is None:
            raise NotImplementedError("please input Event")
        (_, position_ids, attention_mask, _, inputs_embeds, _) = \
            self.prepare_inputs_labels_for_multimodal(
                inputs, position_ids, attention_mask, None, None, event_tensors=event_tensors,
                event_image_sizes=event_image_sizes, event_features=event_features,
                already_adapted=already_adapted)
        return self.generate_from_embeds(inputs_embeds, attention_mask, **kwargs)

    def generate_from_embeds(self, inputs_embeds, attention_mask=None, do_sample=False,
                             temperature=1.0, top_p=1.0, top_k=50, num_beams=1, max_new_tokens=512,
                             use_cache=True, output_hidden_states=False,
                             return_dict_in_generate=False, eos_token_id=None,
                             generator: Optional[torch.Generator] = None,
                             use_engine: Optional[bool] = None, **unused):
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        if not do_sample:
            temperature = 0.0
        if num_beams > 1:
            from ..runtime.beam import beam_search
            seq = beam_search(self, inputs_embeds, attention_mask, num_beams, max_new_tokens, eos)
            return GenerateOutput(seq) if return_dict_in_generate else seq
        bsz = inputs_embeds.shape[0]
        engine_ok = (inputs_embeds.is_cuda and bsz == 1 and not output_hidden_states
                     and inputs_embeds.dtype == torch.bfloat16
                     and (attention_mask is None or bool(attention_mask.all())))
        if use_engine is None:
            use_engine = engine_ok
        if use_engine:
            if not engine_ok:
                raise ValueError("engine path needs a bf16 GPU batch of one without padding")
            seq = self.engine().generate(inputs_embeds[0], max_new_tokens, temperature, top_p,
                                         eos, generator, top_k or 0)[None]
            return GenerateOutput(seq) if return_dict_in_generate else seq

        # module path: dynamic caches, any batch, optional hidden-state capture
        device = inputs_embeds.device
        lengths = attention_mask.long().sum(1) if attention_mask is not None else \
            torch.full((bsz,), inputs_embeds.shape[1], device=device, dtype=torch.long)
        out = self.model(inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                         use_cache=True, output_hidden_states=output_hidden_states)
        last = out.last_hidden_state[torch.arange(bsz, device=device), lengths - 1]
        steps_h = [out.hidden_states] if output_hidden_states else None
        past = out.past_key_values
        mask = attention_mask.bool() if attention_mask is not None else \
            torch.ones(bsz, inputs_embeds.shape[1], dtype=torch.bool, device=device)
        done = torch.zeros(bsz, dtype=torch.bool, device=device)
        tokens = []
        for step in range(max_new_tokens):
            logits = self.lm_head(last)
            nxt = ops.sample_top_p(logits, temperature, top_p, generator, top_k or 0) if temperature > 0 \
                else ops.argmax_last(logits)
            nxt = torch.where(done, torch.full_like(nxt, self.config.pad_token_id or 0), nxt)
            tokens.append(nxt)
            done = done | (nxt == eos)
            if bool(done.all()) or step == max_new_tokens - 1:
                break
            mask = torch.cat([mask, torch.ones(bsz, 1, dtype=torch.bool, device=device)], dim=1)
            pos = (lengths + step)[:, None]
            out = self.model(inputs_embeds=self.model.embed_tokens(nxt)[:, None],
                             attention_mask=mask, position_ids=pos, past_key_values=past,
                             use_cache=True, output_hidden_states=output_hidden_states)
            past = out.past_key_values
            last = out.last_hidden_state[:, -1]
            if output_hidden_states:
                steps_h.append(out.hidden_states)
        seq = torch.stack(tokens, dim=1)
        if return_dict_in_generate:
            return GenerateOutput(seq, tuple(steps_h) if steps_h else None)
        return seq

    # ------------------------------------------------------------------- engine
    def _vision_engine_fits(self) -> bool:
        """The captured tower takes the kernels' geometries (width a multiple of 64, heads of 64 or 128);
        a tower outside them -- the small stand-ins of tests and CPU stubs moved to a GPU -- runs the
        module path, whose ops pick per call."""
        vc = self.config.vision_config
        hd = vc.hidden_size // vc.num_attention_heads
        return vc.hidden_size % 64 == 0 and hd in (64, 128)

    def vision_engine(self, **kwargs):
        """Graph-captured event tower bound to this model's weights."""
        if getattr(self, "_vision_engine", None) is None:
            from ..runtime.vision_engine import VisionEngine
            self._vision_engine = VisionEngine(self, **kwargs)
        return self._vision_engine

    def engine(self, **kwargs):
        """The serving engine bound to this model's weights (built on first use)."""
        if self._engine is None:
            from ..runtime.engine import GenerationEngine
            # engine_defaults: set by eventgpt_amd.quant.apply_quant_args (4-bit weights)
            self._engine = GenerationEngine(self, **{**getattr(self, "engine_defaults", {}), **kwargs})
        return self._engine
