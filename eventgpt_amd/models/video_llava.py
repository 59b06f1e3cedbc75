"""Video-LLaVA-shaped verifier: CLIP-L/14-224 video tower -> 2-layer projector -> LLaMA.

The reference uses HF ``VideoLlavaForConditionalGeneration`` (NF4) as the
speculative-decoding *target* (pipeline/feature_extraction/extract_hidden_states.py:117-165,
pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:197-261, 292-360): 8 uniformly
sampled frames, every frame contributes all 257 tokens of the second-to-last
tower layer, prompt ``USER: <video>\\n{query}\\nASSISTANT:``.  This module keeps
the HF state-dict layout (``video_tower.vision_model.*``,
``multi_modal_projector.linear_{1,2}.*``, ``language_model.model.*``,
``language_model.lm_head.weight``) so such a checkpoint loads, runs in bf16 on the
same engines as the drafter (no 4-bit path is needed with 288 GB of HBM), and
random-initialises for synthetic benchmarks.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from ..utils import checkpoint as ckpt
from .clip_vit import ClipVisionModel
from .config import ClipVisionConfig, EventChatConfig
from .event_chat import GELU, GenerateOutput
from .llama import CausalLMOutput, Linear, LlamaModel

VIDEO_PROMPT = "USER: <video>\n{query}\nASSISTANT:"


class VideoLlavaConfig:
    model_type = "video_llava"

    def __init__(self, text_config: Optional[EventChatConfig] = None,
                 vision_config: Optional[ClipVisionConfig] = None, num_frames: int = 8,
                 vision_feature_layer: int = -2, image_token_index: int = 32000,
                 video_token_index: int = 32001, torch_dtype: str = "bfloat16"):
        self.text_config = text_config or EventChatConfig(vocab_size=32064, torch_dtype=torch_dtype)
        self.vision_config = vision_config or ClipVisionConfig(image_size=224, patch_size=14)
        self.num_frames = num_frames
        self.vision_feature_layer = vision_feature_layer
        self.image_token_index = image_token_index
        self.video_token_index = video_token_index
        self.torch_dtype = torch_dtype

    @property
    def tokens_per_frame(self) -> int:
        return self.vision_config.num_tokens

    @property
    def video_tokens(self) -> int:
        return self.num_frames * self.tokens_per_frame

    def to_dict(self) -> dict:
        t = self.text_config.to_dict()
        t.pop("architectures", None)
        return {"model_type": self.model_type, "text_config": t, "vision_config": self.vision_config.to_dict(),
                "num_frames": self.num_frames, "vision_feature_layer": self.vision_feature_layer,
                "image_token_index": self.image_token_index, "video_token_index": self.video_token_index,
                "torch_dtype": self.torch_dtype, "architectures": ["VideoLlavaForConditionalGeneration"]}

    @classmethod
    def from_dict(cls, d: dict) -> "VideoLlavaConfig":
        vc = {k: v for k, v in d.get("vision_config", {}).items() if k != "model_type"}
        known = ClipVisionConfig().to_dict().keys()
        return cls(EventChatConfig.from_dict(d.get("text_config", {})),
                   ClipVisionConfig(**{k: v for k, v in vc.items() if k in known}) if vc else None,
                   d.get("num_frames", 8), d.get("vision_feature_layer", -2),
                   d.get("image_token_index", 32000), d.get("video_token_index", 32001),
                   d.get("torch_dtype", "bfloat16"))

    @classmethod
    def from_pretrained(cls, path: str) -> "VideoLlavaConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def video_llava_7b(cls) -> "VideoLlavaConfig":
        return cls()

    @classmethod
    def tiny(cls, vocab_size: int = 320, hidden_size: int = 64, layers: int = 2, heads: int = 4,
             num_frames: int = 2, image_size: int = 28) -> "VideoLlavaConfig":
        text = EventChatConfig(vocab_size=vocab_size, hidden_size=hidden_size,
                               intermediate_size=hidden_size * 2 + 32, num_hidden_layers=layers,
                               num_attention_heads=heads, max_position_embeddings=512, torch_dtype="float32")
        return cls(text, ClipVisionConfig.tiny(image_size=image_size), num_frames, -2,
                   vocab_size - 2, vocab_size - 1, "float32")


class VideoLlavaProjector(nn.Module):
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.linear_1 = Linear(in_dim, out_dim)
        self.act = GELU()
        self.linear_2 = Linear(out_dim, out_dim)

    def forward(self, x):
        return self.linear_2(self.linear_1(x, act="gelu"))


class LlamaForCausalLM(nn.Module):
    """``model`` + ``lm_head`` with the attribute surface the serving engine binds to."""

    def __init__(self, config: EventChatConfig):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config)
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False)
        self._engine = None

    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def forward(self, input_ids=None, inputs_embeds=None, attention_mask=None, position_ids=None,
                past_key_values=None, use_cache=False, output_hidden_states=False):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                         past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
                         output_hidden_states=output_hidden_states)
        return CausalLMOutput(self.lm_head(out.last_hidden_state), out.past_key_values, out.hidden_states, None)

    def engine(self, **kw):
        if self._engine is None:
            from ..runtime.engine import GenerationEngine
            self._engine = GenerationEngine(self, **{**getattr(self, "engine_defaults", {}), **kw})
        return self._engine


class VideoLlavaModel(nn.Module):
    config_class = VideoLlavaConfig

    def __init__(self, config: VideoLlavaConfig):
        super().__init__()
        self.config = config
        self.video_tower = ClipVisionModel(config.vision_config)
        self.multi_modal_projector = VideoLlavaProjector(config.vision_config.hidden_size,
                                                         config.text_config.hidden_size)
        self.language_model = LlamaForCausalLM(config.text_config)
        self.video_tower.requires_grad_(False)
        self._vision_engine = None

    # ------------------------------------------------------------ construction
    @classmethod
    def from_config(cls, config: VideoLlavaConfig, dtype=None, device=None, seed: Optional[int] = 0):
        dtype = dtype or getattr(torch, str(config.torch_dtype))
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
    def from_pretrained(cls, path: str, torch_dtype=None, device=None) -> "VideoLlavaModel":
        model = cls.from_config(VideoLlavaConfig.from_pretrained(path), torch_dtype, device, seed=None)
        missing, unexpected = ckpt.load_into(model, path, strict=False)
        model.load_report = {"missing": missing, "unexpected": unexpected}
        return model.eval()

    def save_pretrained(self, path: str, max_shard_bytes: int = 5 << 30) -> None:
        self.config.save_pretrained(path)
        ckpt.save_sharded(self.state_dict(), path, max_shard_bytes)

    # --------------------------------------------------------------- accessors
    @property
    def device(self):
        return self.language_model.device

    @property
    def dtype(self):
        return self.language_model.dtype

    @property
    def lm_head(self):
        return self.language_model.lm_head

    @property
    def model(self):
        return self.language_model.model

    def get_vision_tower(self):
        return self.video_tower

    def get_input_embeddings(self):
        return self.language_model.model.embed_tokens

    def engine(self, **kw):
        return self.language_model.engine(**kw)

    def vision_engine(self, **kw):
        if self._vision_engine is None:
            from ..runtime.vision_engine import VisionEngine
            self._vision_engine = VisionEngine(self, tower=self.video_tower,
                                               projector=[self.multi_modal_projector.linear_1,
                                                          self.multi_modal_projector.linear_2],
                                               adaptor=None, feature_layer=self.config.vision_feature_layer, **kw)
        return self._vision_engine

    # ------------------------------------------------------------------ vision
    def get_video_features(self, pixel_values_videos: torch.Tensor) -> torch.Tensor:
        """``[B,T,3,H,W]`` -> ``[B, T*tokens_per_frame, hidden]``; every token of the
        selected tower layer is kept (CLS included), frames run as one batch."""
        b, t = pixel_values_videos.shape[:2]
        px = pixel_values_videos.flatten(0, 1).to(self.device)
        if px.is_cuda and self.dtype == torch.bfloat16 and not torch.is_grad_enabled():
            f = self.vision_engine().encode(px, pool=False)
        else:
            with torch.no_grad():
                hs = self.video_tower(px.to(self.dtype), output_hidden_states=True).hidden_states
            f = self.multi_modal_projector(hs[self.config.vision_feature_layer])
        return f.reshape(b, t * f.shape[1], f.shape[2])

    def merge_video_features(self, input_ids: torch.Tensor, video_features: torch.Tensor,
                             attention_mask: Optional[torch.Tensor] = None):
        """Replace ``<video>`` placeholders by feature rows.  Accepts either one
        placeholder per video (expanded here) or a pre-expanded run of
        ``video_tokens`` placeholders (scattered in place).  Batch rows must have
        equal lengths after the merge (the benchmarks use batch 1)."""
        vid = self.config.video_token_index
        embed = self.get_input_embeddings()
        rows = []
        for b in range(input_ids.shape[0]):
            ids = input_ids[b]
            if attention_mask is not None:
                ids = ids[attention_mask[b].bool()]
            is_vid = ids == vid
            n = int(is_vid.sum())
            feats = video_features[b].to(embed.weight.dtype)
            text = embed(ids.masked_fill(is_vid, 0))
            if n == feats.shape[0]:
                text = text.clone()
                text[is_vid] = feats
                rows.append(text)
            elif n == 1:
                at = int(torch.nonzero(is_vid)[0])
                rows.append(torch.cat([text[:at], feats, text[at + 1:]]))
            elif n == 0:
                rows.append(text)
            else:
                raise ValueError(f"{n} <video> placeholders for {feats.shape[0]} feature rows")
        if len({r.shape[0] for r in rows}) != 1:
            raise ValueError("merged rows differ in length; pad the batch to equal text lengths")
        return torch.stack(rows)

    # ----------------------------------------------------------------- forward
    def forward(self, input_ids=None, attention_mask=None, pixel_values_videos=None, inputs_embeds=None,
                past_key_values=None, position_ids=None, use_cache=False, output_hidden_states=False):
        if inputs_embeds is None:
            if pixel_values_videos is not None and past_key_values is None:
                feats = self.get_video_features(pixel_values_videos)
                inputs_embeds = self.merge_video_features(input_ids, feats, attention_mask)
                attention_mask = None
            else:
                inputs_embeds = self.get_input_embeddings()(input_ids)
        return self.language_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                                   position_ids=position_ids, past_key_values=past_key_values,
                                   use_cache=use_cache, output_hidden_states=output_hidden_states)

    @torch.no_grad()
    def prepare_embeds(self, input_ids: torch.Tensor, pixel_values_videos: torch.Tensor) -> torch.Tensor:
        return self.merge_video_features(input_ids, self.get_video_features(pixel_values_videos))

    @torch.no_grad()
    def generate(self, input_ids=None, pixel_values_videos=None, attention_mask=None, max_new_tokens: int = 50,
                 do_sample: bool = False, temperature: float = 1.0, top_p: float = 1.0, eos_token_id=None,
                 top_k: int = 50,
                 output_hidden_states: bool = False, return_dict_in_generate: bool = False,
                 use_engine: Optional[bool] = None, generator=None, **unused):
        """Returns only the generated ids ``[B, n_new]`` (prompt goes in as embeddings)."""
        embeds = self.prepare_embeds(input_ids, pixel_values_videos)
        eos = self.config.text_config.eos_token_id if eos_token_id is None else eos_token_id
        temp = temperature if do_sample else 0.0
        engine_ok = embeds.is_cuda and embeds.shape[0] == 1 and embeds.dtype == torch.bfloat16 \
            and not output_hidden_states
        if use_engine is None:
            use_engine = engine_ok
        if use_engine:
            seq = self.engine().generate(embeds[0], max_new_tokens, temp, top_p, eos, generator, top_k or 0)[None]
            return GenerateOutput(seq) if return_dict_in_generate else seq
        lm = self.language_model
        out = lm.model(inputs_embeds=embeds, use_cache=True, output_hidden_states=output_hidden_states)
        past, last = out.past_key_values, out.last_hidden_state[:, -1]
        steps_h = [out.hidden_states] if output_hidden_states else None
        pos = embeds.shape[1]
        done = torch.zeros(embeds.shape[0], dtype=torch.bool, device=embeds.device)
        toks = []
        from .. import ops
        for step in range(max_new_tokens):
            logits = lm.lm_head(last)
            nxt = ops.sample_top_p(logits, temp, top_p, generator, top_k or 0) if temp > 0 else ops.argmax_last(logits)
            nxt = torch.where(done, torch.zeros_like(nxt), nxt)
            toks.append(nxt)
            done = done | (nxt == eos)
            if bool(done.all()) or step == max_new_tokens - 1:
                break
            out = lm.model(inputs_embeds=lm.model.embed_tokens(nxt)[:, None],
                           position_ids=torch.full((embeds.shape[0], 1), pos, device=embeds.device),
                           past_key_values=past, use_cache=True, output_hidden_states=output_hidden_states)
            past, last, pos = out.past_key_values, out.last_hidden_state[:, -1], pos + 1
            if output_hidden_states:
                steps_h.append(out.hidden_states)
        seq = torch.stack(toks, 1)
        if return_dict_in_generate:
            return GenerateOutput(seq, tuple(steps_h) if steps_h else None)
        return seq


# ------------------------------------------------------------------ processing
def sample_frame_indices(total: int, num_frames: int = 8) -> List[int]:
    """Uniform sampling as in the reference's MP4 loader
    (extract_hidden_states.py:84-106): ``linspace(0, total-1, num_frames)``; short
    clips repeat their last frame."""
    if total <= 0:
        return []
    if total >= num_frames:
        return np.linspace(0, total - 1, num_frames, dtype=int).tolist()
    return list(range(total)) + [total - 1] * (num_frames - total)


def load_video_frames(path: str, num_frames: int = 8) -> Optional[List[np.ndarray]]:
    """Frames of a clip as ``[H,W,3]`` uint8 arrays.  ``path`` may be a directory of
    images, an ``.npy``/``.npz`` stack ``[T,H,W,3]``, or a video container (needs
    PyAV, which is optional: without it containers return ``None``)."""
    if os.path.isdir(path):
        from ..data.events import load_image
        names = sorted(n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
        idx = sample_frame_indices(len(names), num_frames)
        return [np.asarray(load_image(os.path.join(path, names[i]))) for i in idx] or None
    if path.endswith((".npy", ".npz")):
        arr = np.load(path)
        arr = arr[arr.files[0]] if hasattr(arr, "files") else arr
        return [np.asarray(arr[i]) for i in sample_frame_indices(arr.shape[0], num_frames)] or None
    try:
        import av
    except ImportError:
        return None
    try:
        with av.open(path) as container:
            frames = [f.to_ndarray(format="rgb24") for f in container.decode(video=0)]
    except Exception:
        return None
    return [frames[i] for i in sample_frame_indices(len(frames), num_frames)] or None


class VideoProcessor:
    """Frames -> ``pixel_values_videos [1,T,3,S,S]`` and prompt -> ids with the
    ``<video>`` placeholder expanded to one id (the model expands it to
    ``T*tokens_per_frame`` rows)."""

    def __init__(self, config: VideoLlavaConfig, tokenizer=None):
        from ..data.clip_preprocess import EventImageProcessor
        self.config = config
        self.tokenizer = tokenizer
        self.image_processor = EventImageProcessor(config.vision_config.image_size)

    def frames(self, frames: Sequence[np.ndarray], device=None, dtype=torch.bfloat16) -> torch.Tensor:
        """``device`` given: preprocess there (one HIP kernel on a GPU) and return ``dtype``;
        otherwise fp32 on the host, like an HF processor."""
        if device is not None:
            from ..ops.event_ops import preprocess_frames
            return preprocess_frames(frames, device, self.config.vision_config.image_size, dtype)[None]
        px = [self.image_processor(np.asarray(f), return_tensors="pt")["pixel_values"][0] for f in frames]
        return torch.stack(px)[None]

    def prompt_ids(self, query: str) -> torch.Tensor:
        text = VIDEO_PROMPT.format(query=query)
        head, tail = text.split("<video>")
        tok = self.tokenizer
        a = tok(head).input_ids
        b = tok(tail).input_ids
        bos = getattr(tok, "bos_token_id", None)
        if bos is not None and b and b[0] == bos:
            b = b[1:]
        return torch.tensor([a + [self.config.video_token_index] + b], dtype=torch.long)

    def __call__(self, text: str, videos: Sequence[np.ndarray], return_tensors: str = "pt", device=None,
                 dtype=torch.bfloat16) -> dict:
        ids = self.prompt_ids(text) if "<video>" not in text else self._raw(text)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids),
                "pixel_values_videos": self.frames(videos, device, dtype)}

    def _raw(self, text: str) -> torch.Tensor:
        head, tail = text.split("<video>")
        tok = self.tokenizer
        a, b = tok(head).input_ids, tok(tail).input_ids
        bos = getattr(tok, "bos_token_id", None)
        if bos is not None and b and b[0] == bos:
            b = b[1:]
        return torch.tensor([a + [self.config.video_token_index] + b], dtype=torch.long)
