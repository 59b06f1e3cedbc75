"""Causal-LM pre-training on tokenised event streams (the ``script/train.py`` slot).

The reference ships an unrelated HF-``Trainer`` template here: a 12x768 GPT-2 trained
with the inputs as their own labels on raw event arrays from ``event_npy/{train,val}``,
warm-up + decay, gradient accumulation, evaluation / checkpoint / logging cadence, early
stopping (script/train.py:26-270).  This is the same job on this package's stack: the
framework's LLaMA decoder, an event tokenizer (polarity x coarse position x time-bin ->
one id), data-parallel ranks over RCCL with the sharded AdamW, the warm-up/cosine
schedule of ``utils.schedules`` and resumable checkpoints.  Data is a directory of event
``.npy`` files (or synthetic streams with ``--synthetic N``).
"""
from __future__ import annotations

import argparse
import json
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from ..data.events import load_event_npy, synthetic_events
from ..models.config import EventChatConfig
from ..models.video_llava import LlamaForCausalLM
from ..parallel.dist import all_reduce_scalar, init_distributed
from ..parallel.dp import ShardedAdamW, broadcast_parameters
from ..utils.schedules import LinearWarmupCosineLR

MODEL_CONFIG = dict(hidden_size=768, intermediate_size=2048, num_hidden_layers=12, num_attention_heads=12,
                    max_position_embeddings=1024)
GRID, TIME_BINS = 16, 8            # 16x16 position cells, 8 inter-event time buckets, 2 polarities
VOCAB = 3 + 2 * GRID * GRID * TIME_BINS


def tokenize_events(ev: Dict[str, np.ndarray], width: Optional[int] = None, height: Optional[int] = None) -> np.ndarray:
    """One token per event: ``3 + ((p * TIME_BINS + dt_bucket) * GRID + cy) * GRID + cx`` where
    ``dt_bucket`` is the log2 bucket of the gap to the previous event."""
    x, y, p = ev["x"].astype(np.int64), ev["y"].astype(np.int64), ev["p"].astype(np.int64)
    t = ev["t"].astype(np.int64)
    w, h = width or int(x.max()) + 1, height or int(y.max()) + 1
    cx, cy = np.minimum(x * GRID // w, GRID - 1), np.minimum(y * GRID // h, GRID - 1)
    dt = np.diff(t, prepend=t[:1])
    bucket = np.minimum(np.floor(np.log2(np.maximum(dt, 0) + 1)).astype(np.int64), TIME_BINS - 1)
    return 3 + ((np.minimum(p, 1) * TIME_BINS + bucket) * GRID + cy) * GRID + cx


def load_token_blocks(files: List[str], block: int, synthetic: int = 0, seed: int = 0) -> torch.Tensor:
    streams = [tokenize_events(synthetic_events(20000, 320, 240, seed=seed + i)) for i in range(synthetic)]
    streams += [tokenize_events(load_event_npy(f, trusted=True)) for f in files]
    ids = np.concatenate(streams) if streams else np.zeros(0, dtype=np.int64)
    n = ids.size // block
    return torch.from_numpy(ids[: n * block].reshape(n, block))


def evaluate(model, blocks: torch.Tensor, batch: int, device) -> float:
    model.eval()
    tot, n = 0.0, 0
    with torch.no_grad():
        for i in range(0, blocks.shape[0], batch):
            b = blocks[i:i + batch].to(device)
            logits = model(input_ids=b).logits
            tot += float(torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, logits.shape[-1]),
                                                           b[:, 1:].reshape(-1), reduction="sum"))
            n += b[:, 1:].numel()
    return tot / max(n, 1)


def main(argv=None) -> Dict:
    p = argparse.ArgumentParser(description="Pre-train a causal LM on tokenised event streams")
    p.add_argument("--data_dir", default=None, help="holds event_npy/train and event_npy/val")
    p.add_argument("--synthetic", type=int, default=0, help="number of synthetic streams per split")
    p.add_argument("--output_dir", default="./eventgpt_lm_output")
    p.add_argument("--block_size", type=int, default=256)
    p.add_argument("--num_train_epochs", type=int, default=10)
    p.add_argument("--per_device_train_batch_size", type=int, default=8)
    p.add_argument("--gradient_accumulation_steps", type=int, default=4)
    p.add_argument("--learning_rate", type=float, default=5e-5)
    p.add_argument("--warmup_steps", type=int, default=500)
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--eval_steps", type=int, default=500)
    p.add_argument("--save_steps", type=int, default=1000)
    p.add_argument("--logging_steps", type=int, default=100)
    p.add_argument("--early_stopping_patience", type=int, default=3)
    p.add_argument("--layers", type=int, default=MODEL_CONFIG["num_hidden_layers"])
    p.add_argument("--hidden", type=int, default=MODEL_CONFIG["hidden_size"])
    p.add_argument("--resume", action="store_true")
    p.add_argument("--cpu", action="store_true")
    args = p.parse_args(argv)

    info = init_distributed(use_gpu=False if args.cpu else None)
    dtype = torch.bfloat16 if info.device.type == "cuda" else torch.float32

    def split(name):
        d = os.path.join(args.data_dir, "event_npy", name) if args.data_dir else None
        files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".npy")) if d and os.path.isdir(d) else []
        return load_token_blocks(files, args.block_size, args.synthetic, seed=0 if name == "train" else 1000)
    train, val = split("train"), split("val")
    if train.shape[0] == 0:
        raise SystemExit("no training data: pass --data_dir or --synthetic N")
    train = train[info.rank::info.world]
    cfg = EventChatConfig(vocab_size=VOCAB, hidden_size=args.hidden,
                          intermediate_size=MODEL_CONFIG["intermediate_size"] * args.hidden // 768,
                          num_hidden_layers=args.layers, num_attention_heads=max(args.hidden // 64, 1),
                          max_position_embeddings=max(args.block_size, 1024), torch_dtype=str(dtype).split(".")[-1])
    torch.manual_seed(0)
    with torch.device(info.device):
        model = LlamaForCausalLM(cfg)
    for prm in model.parameters():
        if prm.ndim >= 2:
            torch.nn.init.normal_(prm, std=0.02)
    model.to(dtype)
    if info.world > 1:
        broadcast_parameters(model)
    opt = ShardedAdamW(model.parameters(), lr=args.learning_rate, weight_decay=args.weight_decay, overlap=False)
    steps_per_epoch = max(1, math.ceil(train.shape[0] / args.per_device_train_batch_size
                                       / args.gradient_accumulation_steps))
    total = steps_per_epoch * args.num_train_epochs
    sched = LinearWarmupCosineLR(opt, total, min_lr=0.0, init_lr=args.learning_rate, warmup_steps=args.warmup_steps,
                                 warmup_start_lr=0.0)
    os.makedirs(args.output_dir, exist_ok=True)
    state_path = os.path.join(args.output_dir, f"trainer_state_rank{info.rank}.pt")
    step, best, patience, log = 0, float("inf"), 0, []
    if args.resume and os.path.exists(state_path):
        st = torch.load(state_path, map_location="cpu", weights_only=False)     # written by this script
        model.load_state_dict(st["model"])
        opt.load_state_dict(st["opt"])
        step, best, patience, log = st["step"], st["best"], st["patience"], st["log"]
    gen = torch.Generator().manual_seed(1234)
    stop = False
    for epoch in range(step // steps_per_epoch, args.num_train_epochs):
        perm = torch.randperm(train.shape[0], generator=gen)
        model.train()
        micro, run_loss = 0, 0.0
        opt.zero_grad()
        for i in range(0, perm.numel(), args.per_device_train_batch_size):
            b = train[perm[i:i + args.per_device_train_batch_size]].to(info.device)
            logits = model(input_ids=b).logits
            loss = torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, logits.shape[-1]),
                                                     b[:, 1:].reshape(-1))
            (loss / args.gradient_accumulation_steps).backward()
            run_loss += float(loss.detach())
            micro += 1
            if micro % args.gradient_accumulation_steps and i + args.per_device_train_batch_size < perm.numel():
                continue
            # per-step schedule: linear warm-up, then a cosine over the optimizer steps
            sched.step(0 if step < args.warmup_steps else step, step)
            opt.step(max_grad_norm=1.0)
            opt.zero_grad()
            step += 1
            if step % args.logging_steps == 0 and info.is_main:
                print(f"[train_lm] step {step}/{total} loss {run_loss / micro:.4f} "
                      f"lr {opt.param_groups[0]['lr']:.2e}", flush=True)
            if val.shape[0] and step % args.eval_steps == 0:
                ev = all_reduce_scalar(evaluate(model, val, args.per_device_train_batch_size, info.device),
                                       device=info.device) / info.world
                log.append({"step": step, "eval_loss": ev, "train_loss": run_loss / micro})
                if ev < best - 1e-4:
                    best, patience = ev, 0
                    if info.is_main:
                        torch.save(model.state_dict(), os.path.join(args.output_dir, "best_model.pt"))
                else:
                    patience += 1
                    stop = patience >= args.early_stopping_patience
                model.train()
            if step % args.save_steps == 0 or stop:
                torch.save({"model": model.state_dict(), "opt": opt.state_dict(), "step": step, "best": best,
                            "patience": patience, "log": log}, state_path)
            if stop:
                break
        if stop:
            break
    final = evaluate(model, val, args.per_device_train_batch_size, info.device) if val.shape[0] else float("nan")
    if info.is_main:
        torch.save(model.state_dict(), os.path.join(args.output_dir, "final_model.pt"))
        with open(os.path.join(args.output_dir, "train_log.json"), "w") as f:
            json.dump({"steps": step, "best_eval_loss": best, "final_eval_loss": final, "log": log,
                       "vocab": VOCAB, "world_size": info.world}, f, indent=2)
    return {"steps": step, "best_eval_loss": best, "final_eval_loss": final, "log": log}


if __name__ == "__main__":
    main()
