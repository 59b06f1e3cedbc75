"""Dump the verifier's LM-head weights for offline token-level metrics.

Output ``vl_lm_head.pt`` = ``{lm_head_weight [V,H] fp32, vocab_size, hidden_dim,
model_id}`` (reference: pipeline/feature_extraction/extract_vl_lm_head.py:26-71).
Only the tensors whose key ends in ``lm_head.weight`` are read from the
checkpoint shards; the 7B model is never instantiated.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict

import torch

from ..utils.checkpoint import iter_checkpoint_tensors


def extract_lm_head(model_path: str) -> torch.Tensor:
    for key, tensor in iter_checkpoint_tensors(model_path):
        if key.endswith("lm_head.weight"):
            return tensor.float().contiguous()
    raise KeyError(f"no lm_head.weight under {model_path}")


def save_lm_head(weight: torch.Tensor, out_path: str, model_id: str) -> Dict:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    blob = {"lm_head_weight": weight.float().cpu().contiguous(), "vocab_size": int(weight.shape[0]),
            "hidden_dim": int(weight.shape[1]), "model_id": model_id}
    torch.save(blob, out_path)
    return blob


def main(argv=None) -> str:
    p = argparse.ArgumentParser(description="Extract the verifier LM head")
    p.add_argument("--model_path", default=None, help="checkpoint directory of the verifier")
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"],
                   help="the randomly initialised verifier of the synthetic flows (same seed as the extractors)")
    p.add_argument("--output", default=None)
    args = p.parse_args(argv)
    out = args.output or os.path.join(os.getcwd(), "data", "vl_lm_head.pt")
    if args.random_init:
        from ..models.video_llava import VideoLlavaConfig, VideoLlavaModel
        cfg = VideoLlavaConfig.video_llava_7b() if args.random_init == "7b" else VideoLlavaConfig.tiny()
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        dtype = torch.bfloat16 if dev == "cuda" else torch.float32
        vl = VideoLlavaModel.from_config(cfg, dtype, dev, seed=1)
        w = vl.language_model.lm_head.weight.detach().float().cpu().contiguous()
        args.model_path = f"random_init:{args.random_init}"
    elif args.model_path:
        w = extract_lm_head(args.model_path)
    else:
        p.error("--model_path or --random_init")
    save_lm_head(w, out, args.model_path)
    print(f"saved {tuple(w.shape)} fp32 ({w.numel() * 4 / 2 ** 20:.1f} MiB) -> {out}")
    return out


if __name__ == "__main__":
    main()
