"""Model configuration objects.

``EventChatConfig`` reads and writes the HF-style ``config.json`` of an
EventGPT checkpoint (``model_type = "EventChat_llama"``, LLaMA fields plus the
multimodal switches ``mm_visual_tower``, ``event_feature_adaptor``,
``use_event_qformer``, ``mm_use_im_start_end``, ``mm_use_im_patch_token`` —
reference: model/EventChatModel.py:41-42,80-90, inference.py:33-34).
It is a plain attribute bag: unknown keys in the JSON are preserved and
round-tripped, and optional switches are *absent* (not ``None``) unless set, so
``hasattr(config, "event_feature_adaptor")`` keeps its meaning.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, Optional


class ClipVisionConfig:
    """Geometry of the CLIP vision transformer used as the event tower."""

    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                 num_attention_heads=16, image_size=336, patch_size=14, num_channels=3,
                 layer_norm_eps=1e-5, hidden_act="quick_gelu", **extra):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.image_size = image_size
        self.patch_size = patch_size
        self.num_channels = num_channels
        self.layer_norm_eps = layer_norm_eps
        self.hidden_act = hidden_act
        for k, v in extra.items():
            setattr(self, k, v)

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def num_tokens(self) -> int:
        return self.num_patches + 1

    def to_dict(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_")}

    @classmethod
    def from_pretrained(cls, path: str) -> "ClipVisionConfig":
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d = d.get("vision_config", d)
        d.pop("model_type", None)
        return cls(**d)

    @classmethod
    def vit_l_14_336(cls) -> "ClipVisionConfig":
        return cls()

    @classmethod
    def tiny(cls, image_size=28, patch_size=14) -> "ClipVisionConfig":
        return cls(hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                   num_attention_heads=4, image_size=image_size, patch_size=patch_size)


class EventChatConfig:
    model_type = "EventChat_llama"

    _DEFAULTS = dict(
        vocab_size=32000, hidden_size=4096, intermediate_size=11008,
        num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=None,
        hidden_act="silu", max_position_embeddings=4096, rms_norm_eps=1e-6,
        rope_theta=10000.0, bos_token_id=1, eos_token_id=2, pad_token_id=0,
        tie_word_embeddings=False, torch_dtype="bfloat16",
    )

    def __init__(self, **kwargs):
        merged = dict(self._DEFAULTS)
        merged.update(kwargs)
        merged.pop("model_type", None)
        for k, v in merged.items():
            setattr(self, k, v)
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads

    # -- derived ------------------------------------------------------------
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    # -- (de)serialisation ----------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        d = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if not k.startswith("_")}
        if isinstance(d.get("vision_config"), ClipVisionConfig):
            d["vision_config"] = d["vision_config"].to_dict()
        d["model_type"] = self.model_type
        d.setdefault("architectures", ["EventChatModel"])
        return d

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "EventChatConfig":
        d = dict(d)
        vc = d.get("vision_config")
        cfg = cls(**d)
        if isinstance(vc, dict):
            cfg.vision_config = ClipVisionConfig(**{k: v for k, v in vc.items()
                                                    if k != "model_type"})
        return cfg

    @classmethod
    def from_pretrained(cls, path: str) -> "EventChatConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f))

    # -- presets ----------------------------------------------------------------
    @classmethod
    def eventgpt_7b(cls, vocab_size: int = 32000) -> "EventChatConfig":
        """Vicuna-7B decoder + CLIP ViT-L/14-336 tower + feature adaptor."""
        cfg = cls(vocab_size=vocab_size, mm_visual_tower="clip-vit-large-patch14-336",
                  event_feature_adaptor=True, mm_use_im_start_end=False,
                  mm_use_im_patch_token=True, max_position_embeddings=4096)
        cfg.vision_config = ClipVisionConfig.vit_l_14_336()
        return cfg

    @classmethod
    def stub_125m(cls, vocab_size: int = 32000) -> "EventChatConfig":
        """OPT-125M-sized decoder (12 layers × 768, 12 heads) with a small tower,
        for CPU plumbing tests."""
        cfg = cls(vocab_size=vocab_size, hidden_size=768, intermediate_size=3072,
                  num_hidden_layers=12, num_attention_heads=12,
                  max_position_embeddings=2048, mm_visual_tower="stub-vit",
                  event_feature_adaptor=True, torch_dtype="float32")
        cfg.vision_config = ClipVisionConfig(hidden_size=64, intermediate_size=128,
                                             num_hidden_layers=2, num_attention_heads=4,
                                             image_size=336, patch_size=14)
        return cfg

    @classmethod
    def tiny(cls, vocab_size: int = 320, hidden_size: int = 64, layers: int = 2,
             heads: int = 4, image_size: int = 28) -> "EventChatConfig":
        cfg = cls(vocab_size=vocab_size, hidden_size=hidden_size,
                  intermediate_size=hidden_size * 2 + 32, num_hidden_layers=layers,
                  num_attention_heads=heads, max_position_embeddings=512,
                  mm_visual_tower="tiny-vit", event_feature_adaptor=True,
                  torch_dtype="float32")
        cfg.vision_config = ClipVisionConfig.tiny(image_size=image_size)
        return cfg
