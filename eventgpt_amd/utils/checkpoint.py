"""HF-layout checkpoint IO (safetensors / sharded ``.bin``), no HF runtime needed.

A checkpoint directory holds ``config.json`` plus either
``model.safetensors`` / ``model-0000x-of-0000y.safetensors`` +
``model.safetensors.index.json`` or the ``pytorch_model*.bin`` equivalents.
``.bin`` shards are read with ``torch.load(weights_only=True)`` only.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterator, List, Optional, Tuple

import torch

SAFE_INDEX = "model.safetensors.index.json"
BIN_INDEX = "pytorch_model.bin.index.json"


def _shard_files(path: str) -> List[str]:
    for index_name in (SAFE_INDEX, BIN_INDEX):
        idx = os.path.join(path, index_name)
        if os.path.exists(idx):
            with open(idx) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            return [os.path.join(path, s) for s in files]
    for pattern in ("model.safetensors", "model-*.safetensors", "pytorch_model.bin",
                    "pytorch_model-*.bin"):
        found = sorted(glob.glob(os.path.join(path, pattern)))
        if found:
            return found
    raise FileNotFoundError(f"no weight files under {path}")


def iter_checkpoint_tensors(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield ``(key, tensor)`` shard by shard so peak host memory is one shard."""
    for shard in _shard_files(path):
        if shard.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(shard, framework="pt", device="cpu") as f:
                for k in f.keys():
                    yield k, f.get_tensor(k)
        else:
            sd = torch.load(shard, map_location="cpu", weights_only=True, mmap=True)
            for k, v in sd.items():
                yield k, v
            del sd


def load_into(model: torch.nn.Module, path: str, strict: bool = True,
              ignore_suffixes: Tuple[str, ...] = ("position_ids", "rotary_emb.inv_freq")
              ) -> Tuple[List[str], List[str]]:
    """Copy checkpoint tensors into ``model`` in place (dtype/device of the model).

    Returns ``(missing, unexpected)``.  Rows are copied into a *prefix* of a
    larger parameter when the only mismatch is a grown vocabulary.
    """
    target = dict(model.state_dict(keep_vars=True))
    seen = set()
    unexpected: List[str] = []
    with torch.no_grad():
        for k, v in iter_checkpoint_tensors(path):
            if k.endswith(ignore_suffixes):
                continue
            t = target.get(k)
            if t is None:
                unexpected.append(k)
                continue
            if t.shape == v.shape:
                t.copy_(v)
            elif t.ndim == v.ndim and t.shape[1:] == v.shape[1:] and t.shape[0] > v.shape[0]:
                t[: v.shape[0]].copy_(v)
            else:
                raise ValueError(f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(t.shape)}")
            seen.add(k)
    missing = [k for k in target if k not in seen]
    if strict and (missing or unexpected):
        raise KeyError(f"missing={missing[:8]}… unexpected={unexpected[:8]}…")
    return missing, unexpected


def save_sharded(state_dict: Dict[str, torch.Tensor], path: str,
                 max_shard_bytes: int = 5 << 30, metadata: Optional[dict] = None) -> List[str]:
    """Write ``state_dict`` as safetensors shards + index (HF naming)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    shards: List[Dict[str, torch.Tensor]] = [{}]
    size = 0
    for k, v in state_dict.items():
        nbytes = v.numel() * v.element_size()
        if shards[-1] and size + nbytes > max_shard_bytes:
            shards.append({})
            size = 0
        shards[-1][k] = v.detach().to("cpu").contiguous().clone()
        size += nbytes
    names = []
    meta = {"format": "pt"}
    if len(shards) == 1:
        names = ["model.safetensors"]
        save_file(shards[0], os.path.join(path, names[0]), metadata=meta)
    else:
        weight_map = {}
        total = 0
        for i, sh in enumerate(shards):
            name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            names.append(name)
            save_file(sh, os.path.join(path, name), metadata=meta)
            for k, v in sh.items():
                weight_map[k] = name
                total += v.numel() * v.element_size()
        with open(os.path.join(path, SAFE_INDEX), "w") as f:
            json.dump({"metadata": {"total_size": total, **(metadata or {})},
                       "weight_map": weight_map}, f, indent=2)
    return names
