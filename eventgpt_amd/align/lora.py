"""Low-rank adapters on the decoder's attention projections (the "L6" level).

The reference wraps an NF4-quantised EventGPT with PEFT LoRA (r=16, alpha=32,
dropout 0.05 on q/k/v/o — pipeline/adapter_train/train_lora_adapter.py:335-402).
Here the base weights stay bf16 (a 7B model is 5 % of one MI355X's HBM, so
4-bit storage buys nothing) and LoRA is a small module of this package:

    y = W x + (alpha / r) * B (A dropout(x))         A: [r, in] kaiming,  B: [out, r] zeros

``save_lora`` / ``load_lora`` use PEFT's on-disk layout (``adapter_config.json`` +
``adapter_model.safetensors`` with ``base_model.model.<module>.lora_{A,B}.weight``
keys), so adapters move between the two stacks.  ``merge_lora`` folds the update
into the base weights for serving on the graph-captured engine.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Iterable, List, Optional, Sequence

import torch
from torch import nn

DEFAULT_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj")


class LoRALinear(nn.Module):
    """Wraps a frozen linear layer; same call signature as ``models.llama.Linear``."""

    def __init__(self, base: nn.Linear, r: int = 16, alpha: int = 32, dropout: float = 0.05):
        super().__init__()
        self.base = base
        self.r, self.alpha, self.scaling = r, alpha, alpha / r
        # the framework's Linear (an nn.Linear whose forward / dgrad / wgrad run the in-tree MFMA
        # GEMM on bf16 GPU tensors): as plain nn.Linear the two low-rank products of every wrapped
        # layer -- and their backward -- were 128 library GEMM launches per step at 7B
        from ..models.llama import Linear
        self.lora_A = Linear(base.in_features, r, bias=False, device=base.weight.device, dtype=base.weight.dtype)
        self.lora_B = Linear(r, base.out_features, bias=False, device=base.weight.device, dtype=base.weight.dtype)
        nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B.weight)
        self.lora_dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        base.weight.requires_grad_(False)
        if base.bias is not None:
            base.bias.requires_grad_(False)
        self.merged = False

    @property
    def weight(self):
        return self.base.weight

    @property
    def bias(self):
        return self.base.bias

    @property
    def in_features(self):
        return self.base.in_features

    @property
    def out_features(self):
        return self.base.out_features

    def delta(self) -> torch.Tensor:
        return (self.lora_B.weight.float() @ self.lora_A.weight.float()) * self.scaling

    def lora_delta(self, x):
        """The low-rank update alone, ``(alpha / r) B A dropout(x)``, for a caller that computes the
        base projection itself (several wrapped layers grouped into one GEMM); None once merged."""
        if self.merged:
            return None
        return self._low(x)

    def drop_p(self) -> float:
        """The input dropout probability in effect (0 in eval mode)."""
        d = self.lora_dropout
        return float(d.p) if isinstance(d, nn.Dropout) and d.training else 0.0

    def adapter_entry(self, offset: int):
        """``(column offset, A, B)`` of this adapter for ``ops.lora_projection``."""
        return (offset, self.lora_A.weight, self.lora_B.weight)

    def _low(self, x):
        # the scale rides on the rank-r intermediate ([rows, r]), not on the [rows, out] product
        from ..ops import lowrank_delta
        return lowrank_delta(self.lora_dropout(x), self.lora_A.weight, self.lora_B.weight, self.scaling)

    def forward(self, x, act: Optional[str] = None, residual=None):
        plain = type(self.base) is nn.Linear
        if self.merged:
            if plain:
                y = self.base(x)
                return y if residual is None else y + residual
            return self.base(x, act=act, residual=residual)
        if act:
            raise NotImplementedError("LoRA on a layer with a fused activation is not supported")
        from .. import ops
        fused = ops.lora_projection(x, self.base.weight, [self.adapter_entry(0)], self.scaling, self.drop_p(),
                                    residual) if self.base.bias is None else None
        if fused is not None:
            return fused
        low = self._low(x)
        extra = low if residual is None else low + residual
        # the low-rank update rides on the base GEMM's residual epilogue
        return self.base(x) + extra if plain else self.base(x, residual=extra)

    @torch.no_grad()
    def merge(self) -> None:
        if not self.merged:
            self.base.weight.add_(self.delta().to(self.base.weight.dtype))
            self.merged = True

    @torch.no_grad()
    def unmerge(self) -> None:
        if self.merged:
            self.base.weight.sub_(self.delta().to(self.base.weight.dtype))
            self.merged = False


def _parent(model: nn.Module, name: str):
    parts = name.split(".")
    m = model
    for p in parts[:-1]:
        m = getattr(m, p)
    return m, parts[-1]


def apply_lora(model: nn.Module, r: int = 16, alpha: int = 32, dropout: float = 0.05,
               target_modules: Sequence[str] = DEFAULT_TARGETS, freeze_base: bool = True,
               exclude: Sequence[str] = ("visual_tower",)) -> List[str]:
    """Freeze the model and wrap every ``nn.Linear`` whose leaf name is in
    ``target_modules``; returns the wrapped module paths.  Modules under a path that
    contains an ``exclude`` entry are left alone: the reference's name match also
    wraps the CLIP tower's q/k/v (2.4 M of its 19.1 M LoRA parameters), which never
    train because the tower runs without gradients."""
    if freeze_base:
        for p in model.parameters():
            p.requires_grad_(False)
    wrapped = []
    for name, mod in list(model.named_modules()):
        if any(e in name.split(".") for e in exclude):
            continue
        if isinstance(mod, nn.Linear) and not isinstance(mod, LoRALinear) and name.split(".")[-1] in target_modules:
            parent, leaf = _parent(model, name)
            if isinstance(parent, LoRALinear):
                continue
            setattr(parent, leaf, LoRALinear(mod, r, alpha, dropout))
            wrapped.append(name)
    if not wrapped:
        raise ValueError(f"no module matched {tuple(target_modules)}")
    model._lora_config = {"r": r, "lora_alpha": alpha, "lora_dropout": dropout,
                          "target_modules": list(target_modules)}
    if hasattr(model, "_engine"):
        model._engine = None        # a serving engine built earlier holds the unwrapped weights
    return wrapped


def lora_modules(model: nn.Module) -> Dict[str, LoRALinear]:
    return {n: m for n, m in model.named_modules() if isinstance(m, LoRALinear)}


def lora_parameters(model: nn.Module) -> List[nn.Parameter]:
    return [p for m in lora_modules(model).values() for p in (m.lora_A.weight, m.lora_B.weight)]


def lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, m in lora_modules(model).items():
        sd[f"base_model.model.{name}.lora_A.weight"] = m.lora_A.weight.detach().cpu().contiguous()
        sd[f"base_model.model.{name}.lora_B.weight"] = m.lora_B.weight.detach().cpu().contiguous()
    return sd


def count_parameters(model: nn.Module) -> Dict[str, int]:
    train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    total = sum(p.numel() for p in model.parameters())
    return {"trainable": train, "total": total, "percent": 100.0 * train / max(total, 1)}


def save_lora(model: nn.Module, out_dir: str, base_model_name: str = "") -> str:
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    cfg = dict(getattr(model, "_lora_config", {}))
    cfg.update({"peft_type": "LORA", "task_type": "CAUSAL_LM", "bias": "none",
                "base_model_name_or_path": base_model_name, "inference_mode": True, "fan_in_fan_out": False})
    with open(os.path.join(out_dir, "adapter_config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    save_file(lora_state_dict(model), os.path.join(out_dir, "adapter_model.safetensors"))
    return out_dir


def load_lora(model: nn.Module, adapter_dir: str, trainable: bool = False) -> List[str]:
    """Wrap ``model`` according to the adapter's config and load its weights.  Keys may
    carry PEFT's ``.default`` adapter-name segment."""
    from safetensors.torch import load_file
    with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
        cfg = json.load(f)
    if not lora_modules(model):
        apply_lora(model, cfg.get("r", 16), cfg.get("lora_alpha", 32), cfg.get("lora_dropout", 0.0),
                   cfg.get("target_modules", DEFAULT_TARGETS))
    sd = load_file(os.path.join(adapter_dir, "adapter_model.safetensors"))
    mods = lora_modules(model)
    loaded = []
    for key, t in sd.items():
        k = key.replace("base_model.model.", "", 1).replace(".default", "")
        for which in ("lora_A", "lora_B"):
            suffix = f".{which}.weight"
            if k.endswith(suffix):
                name = k[: -len(suffix)]
                if name not in mods:
                    raise KeyError(f"adapter tensor for unknown module {name}")
                w = getattr(mods[name], which).weight
                w.data.copy_(t.to(w.dtype))
                loaded.append(key)
    for p in lora_parameters(model):
        p.requires_grad_(trainable)
    return loaded


@torch.no_grad()
def merge_lora(model: nn.Module, unwrap: bool = True) -> int:
    """Fold every adapter into its base weight; with ``unwrap`` the wrappers are replaced
    by the plain layers so the serving engine sees an ordinary model."""
    n = 0
    for name, m in list(lora_modules(model).items()):
        m.merge()
        if unwrap:
            parent, leaf = _parent(model, name)
            setattr(parent, leaf, m.base)
        n += 1
    if hasattr(model, "_engine"):
        model._engine = None
    return n
