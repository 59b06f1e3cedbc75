"""``inference.py``-compatible command line (reference: inference.py:11-66).

Same flags and flow: load config / tokenizer / model (bf16 on GPU), add the
``<ev_patch>`` (and optionally ``<ev_start>/<ev_end>``) tokens per the config,
resize embeddings, build the Vicuna-v1 prompt, rasterise + preprocess the event
file into 5 frames, generate, print the decoded answer.

Extras for environments without a checkpoint: ``--random_init {7b,stub125m,tiny}``
builds a random-initialised model instead of reading ``--model_path``, and
``--device/--dtype`` override placement (CPU fp32 is the plumbing configuration).
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Optional, Sequence

import torch

from ..constants import (DEFAULT_EV_END_TOKEN, DEFAULT_EV_START_TOKEN, DEFAULT_EVENT_PATCH_TOKEN,
                         EVENT_TOKEN_INDEX, DEFAULT_NUM_EVENT_FRAMES)
from ..data.conversation import prepare_event_prompt
from ..data.events import event_frames, load_event_npy
from ..data.tokenizer import ByteTokenizer, load_tokenizer, tokenizer_event_token
from ..models import EventChatConfig, EventChatModel


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="EventGPT inference")
    p.add_argument("--model_path", type=str, default=None)
    p.add_argument("--model_base", type=str, default=None)
    p.add_argument("--query", type=str, required=True)
    p.add_argument("--conv_mode", type=str, default="eventgpt_v1")
    p.add_argument("--sep", type=str, default=",")
    p.add_argument("--context_len", type=int, default=2048)
    p.add_argument("--temperature", type=float, default=0.6)
    p.add_argument("--top_p", type=float, default=1)
    p.add_argument("--top_k", type=int, default=50,
                   help="sampling keeps the k most likely tokens (0 = off); 50 is what the reference's "
                        "generate() call inherits from its generation defaults")
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--max_new_tokens", type=int, default=512)
    p.add_argument("--spatial_temporal_encoder", type=bool, default=True)
    p.add_argument("--event_frame", type=str, required=True)
    # extensions
    p.add_argument("--random_init", choices=["7b", "stub125m", "tiny"], default=None)
    from ..quant import add_quant_args
    add_quant_args(p)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--dtype", choices=["bfloat16", "float32"], default=None)
    p.add_argument("--num_frames", type=int, default=DEFAULT_NUM_EVENT_FRAMES)
    p.add_argument("--raster_conflict", choices=["last", "negative"], default="last")
    p.add_argument("--trusted_pickle", action="store_true",
                   help="load the event .npy with numpy's general unpickler")
    p.add_argument("--seed", type=int, default=0)
    return p


def process_event_data(event_frame_path: str, event_processor, device, dtype=torch.bfloat16,
                       n_frames: int = DEFAULT_NUM_EVENT_FRAMES, conflict: str = "last",
                       trusted: bool = False):
    """Event file → ``(image_size [H, W], list of [3, S, S] tensors)``
    (reference: common/common.py:110-129)."""
    ev = load_event_npy(event_frame_path, trusted=trusted)
    frames = event_frames(ev, n_frames, conflict=conflict)
    size = list(frames[0].shape[:2])
    tensors = [event_processor(f, return_tensors="pt")["pixel_values"][0].to(device, dtype=dtype)
               for f in frames]
    return size, tensors


def load_model(args):
    device = torch.device(args.device or ("cuda" if torch.cuda.is_available() else "cpu"))
    dtype = {"bfloat16": torch.bfloat16, "float32": torch.float32}[
        args.dtype or ("bfloat16" if device.type == "cuda" else "float32")]
    if args.random_init:
        cfg = {"7b": EventChatConfig.eventgpt_7b, "stub125m": EventChatConfig.stub_125m,
               "tiny": EventChatConfig.tiny}[args.random_init]()
        tokenizer = ByteTokenizer(cfg.vocab_size)
        model = EventChatModel.from_config(cfg, dtype, device, seed=args.seed)
    else:
        if not args.model_path:
            raise SystemExit("--model_path is required unless --random_init is given")
        cfg = EventChatConfig.from_pretrained(args.model_path)
        tokenizer = load_tokenizer(args.model_path, cfg.vocab_size)
        model = EventChatModel.from_pretrained(args.model_path, torch_dtype=dtype, config=cfg,
                                               device=device)
    return model, tokenizer, device, dtype


def run(argv: Optional[Sequence[str]] = None) -> str:
    args = build_parser().parse_args(argv)
    model, tokenizer, device, dtype = load_model(args)
    if args.use_4bit and device.type != "cuda":
        raise SystemExit("--use_4bit selects the GPU engine's weight format; on the CPU use "
                         "eventgpt_amd.quant.fake_quantize_model_ to study 4-bit quality")
    from ..quant import apply_quant_args
    apply_quant_args(args, model)
    if getattr(model.config, "mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_EVENT_PATCH_TOKEN], special_tokens=True)
    if getattr(model.config, "mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_EV_START_TOKEN, DEFAULT_EV_END_TOKEN], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))
    model.eval()
    if device.type == "cuda" and dtype == torch.bfloat16 and args.num_beams == 1:
        # the serving engine packs its weights and caches now, before the vision tower has allocated
        # and freed anything: decode steps measure 1 % faster on memory handed out this early
        model.engine()

    event_processor = model.get_visual_tower().event_processor
    prompt = prepare_event_prompt(args.query, args.conv_mode)
    size, tensors = process_event_data(args.event_frame, event_processor, device, dtype,
                                       args.num_frames, args.raster_conflict, args.trusted_pickle)
    input_ids = tokenizer_event_token(prompt, tokenizer, EVENT_TOKEN_INDEX,
                                      return_tensors="pt").unsqueeze(0).to(device)
    gen = torch.Generator(device=device).manual_seed(args.seed)
    with torch.inference_mode():
        out = model.generate(input_ids, event_tensors=tensors, event_image_sizes=size,
                             do_sample=args.temperature > 0, temperature=args.temperature,
                             top_p=args.top_p, top_k=args.top_k, num_beams=args.num_beams,
                             max_new_tokens=args.max_new_tokens, use_cache=True, generator=gen)
    text = tokenizer.batch_decode(out, skip_special_tokens=True)[0].strip()
    print(text)
    return text


if __name__ == "__main__":
    run()
