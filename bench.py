#!/usr/bin/env python
"""Headline benchmark: EventGPT-7B bf16 greedy decode tokens/s (+ event-prefill ms).

Matches the reference's headline measurement (decode throughput 54.0 tok/s,
LLM prefill 66.2 ms @ 636 tokens, vision 5.9 ms/frame on an RTX 4090 —
feasible/benchmark_inference/BENCHMARK_1FRAME_VS_8FRAMES_20260124_193405.md:18-29):
EventGPT-7B (Vicuna-7B decoder + CLIP ViT-L/14-336 tower), bf16, batch 1 per
stream, greedy, 1 event frame -> 636-token multimodal prompt.

    python bench.py --gpus N --steps K --warmup W

A *step* is one decode step (one new token per stream).  With N > 1 (launched
under ``python -m torch.distributed.run``) every rank serves its own stream
(data-parallel replicas, weak scaling) and the reported value is the total
tokens/s over all ranks; ``--parallelism tp`` instead shards the decoder
tensor-parallel over the N GPUs for one stream (strong scaling).

Weights are random-initialised with the 7B architecture and the prompt is a
synthetic event stream (no checkpoints or datasets are reachable here).
Rank 0 prints exactly one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

BASELINE_DECODE_TOK_S = 54.0
BASELINE_PREFILL_MS = 66.16
BASELINE_VISION_MS = 5.93


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt-len", type=int, default=636,
                    help="multimodal prompt length (636 = 1 event frame: 578 event + 58 text)")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--max-len", type=int, default=2048, help="KV-cache capacity per stream")
    ap.add_argument("--parallelism", choices=["dp", "tp"], default="dp")
    ap.add_argument("--model", choices=["7b", "small"], default="7b")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--batch", type=int, default=1,
                    help="event streams decoded together per engine (32 = the batched serving config)")
    ap.add_argument("--nsplit", type=int, default=None, help="KV splits of the decode attention")
    ap.add_argument("--weight-format", choices=["bf16", "nf4", "fp4"], default="bf16",
                    help="nf4/fp4: the reference's 4-bit rows (bitsandbytes load_in_4bit); not the headline config")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from eventgpt_amd.models import ClipVisionConfig, EventChatConfig, EventChatModel
    from eventgpt_amd.data import EventImageProcessor, event_frames, synthetic_events

    if args.model == "7b":
        cfg = EventChatConfig.eventgpt_7b(vocab_size=32003)
        model_name = "EventGPT-7B (Vicuna-7B + CLIP ViT-L/14-336, random init)"
    else:
        cfg = EventChatConfig(vocab_size=32003, hidden_size=1024, intermediate_size=2816,
                              num_hidden_layers=4, num_attention_heads=8,
                              mm_visual_tower="small", event_feature_adaptor=True)
        cfg.vision_config = ClipVisionConfig(hidden_size=256, intermediate_size=1024,
                                             num_hidden_layers=2, num_attention_heads=4)
        model_name = "small-debug"
    # tensor-parallel ranks are shards of ONE model and must see identical inputs;
    # data-parallel ranks are independent replicas with their own streams
    data_seed = 0 if (args.parallelism == "tp" and world > 1) else rank
    torch.manual_seed(1234 + data_seed)
    model = EventChatModel.from_config(cfg, torch.bfloat16, dev, seed=1234)

    ekw = {} if args.nsplit is None else {"nsplit": args.nsplit}
    if args.weight_format != "bf16":
        ekw["weight_format"] = args.weight_format
    tp = args.parallelism == "tp" and world > 1
    if tp or args.batch > 1:
        from eventgpt_amd.parallel.tp import TensorParallelEngine
        ekw["max_batch"] = args.batch
        engine = TensorParallelEngine(model, dist.group.WORLD if tp else None, max_len=args.max_len,
                                      max_new_tokens=args.steps + args.warmup + 8,
                                      use_graph=not args.no_graph, **ekw)
    else:
        engine = model.engine(max_len=args.max_len, max_new_tokens=args.steps + args.warmup + 8,
                              use_graph=not args.no_graph, **ekw)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- event prefill: raster -> CLIP preprocess -> ViT -> projector -> pool
    proc = EventImageProcessor(cfg.vision_config.image_size)
    ev = synthetic_events(seed=data_seed)
    frames = event_frames(ev, args.frames)
    tens = [proc(f, return_tensors="pt")["pixel_values"][0].to(dev, torch.bfloat16) for f in frames]
    with torch.no_grad():
        for _ in range(2):
            feats = model.encode_event_frames(tens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_vis = 5
        for _ in range(n_vis):
            feats = model.encode_event_frames(tens)
        torch.cuda.synchronize()
        vision_ms = (time.perf_counter() - t0) * 1e3 / n_vis
    n_event = feats.shape[0]
    n_text = max(args.prompt_len - n_event, 1)
    text_ids = torch.randint(3, 32000, (n_text,), device=dev)
    with torch.no_grad():
        text = model.model.embed_tokens(text_ids)
        embeds = torch.cat([text[:35], feats.to(text.dtype), text[35:]])  # splice position of <event>
    S = embeds.shape[0]

    # ---- LLM prefill latency (one stream)
    def prefill_once():
        engine.reset()
        engine.prefill(embeds, select="greedy" if args.batch == 1 else None)

    for _ in range(2):
        prefill_once()
    torch.cuda.synchronize()
    n_pf = 5
    t0 = time.perf_counter()
    for _ in range(n_pf):
        prefill_once()
    torch.cuda.synchronize()
    prefill_ms = (time.perf_counter() - t0) * 1e3 / n_pf

    # ---- decode: W warm-up steps, then exactly K timed steps
    engine.eos = -1
    engine.reset()
    if args.batch == 1:
        engine.prefill(embeds)
    else:
        engine.prefill_batch([embeds] * args.batch)
    for _ in range(args.warmup):
        engine.decode_step(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        engine.decode_step(True)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    streams = (1 if tp else world) * args.batch
    tok_s = streams * args.steps / dt
    if rank == 0:
        print(json.dumps({
            "metric": "decode tokens/sec (EventGPT-7B bf16 greedy, bs 1 per stream)" if args.batch == 1
            else f"decode tokens/sec (EventGPT-7B bf16 greedy, {args.batch} streams per engine)",
            "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong" if tp else "weak",
            "vs_baseline": round(tok_s / BASELINE_DECODE_TOK_S, 3),
            "dtype": "bf16" if args.weight_format == "bf16" else f"bf16 compute, {args.weight_format} weights",
            "data": "synthetic event stream, random-init weights",
            "event_prefill_ms": round(prefill_ms, 3),
            "event_prefill_vs_baseline": round(BASELINE_PREFILL_MS / prefill_ms, 3),
            "vision_ms_per_frame": round(vision_ms / args.frames, 3),
            "vision_vs_baseline": round(BASELINE_VISION_MS / (vision_ms / args.frames), 3),
            "config": {"model": model_name, "global_batch": streams, "seq_len": S,
                       "event_frames": args.frames,
                       "parallelism": f"{args.parallelism}{world}"},
        }), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
