"""Re-extract only the drafter's hidden states from another event-image set
(``event_image_1f`` by default) and pair them with the verifier states already
stored, chunk by chunk (reference: pipeline/feature_extraction/extract_hidden_states_1f.py:37-406).

Each output chunk records the ``source_chunk`` it was derived from; ``--resume``
skips source chunks that are already present in the output index.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict

import torch

from ..align.store import ChunkedHiddenStateWriter, _load_pt, read_index
from .dataset import InstructionDataset, load_questions
from .extract_hidden_states import MIN_PAIR_LEN, HiddenStateExtractor, StopFlag


def reextract(extractor: HiddenStateExtractor, ds: InstructionDataset, questions, existing_dir: str,
              writer: ChunkedHiddenStateWriter, stop=None) -> Dict[str, int]:
    done = writer.processed_source_chunks()
    stats = {"new": 0, "errors": 0, "skipped_chunks": 0}
    for info in read_index(existing_dir)["chunks"]:
        name = info["path"]
        if name in done:
            stats["skipped_chunks"] += 1
            continue
        if stop is not None and stop.set:
            break
        chunk = _load_pt(os.path.join(existing_dir, "chunks", name))
        kept = 0
        for i, meta in enumerate(chunk["metadata"]):
            si, qi = meta["sample_idx"], meta["question_idx"]
            if si >= len(ds) or qi >= len(questions):
                stats["errors"] += 1
                continue
            res = extractor.extract_egpt_hidden(ds, ds[si], questions[qi])
            n = min(res[0].shape[0], int(chunk["seq_lens"][i])) if res is not None else 0
            if n < MIN_PAIR_LEN:
                stats["errors"] += 1
                continue
            new_meta = dict(meta, egpt_text=res[2], seq_len=n, event_image_key=extractor.event_image_key)
            writer.add_sample(res[0][:n], chunk["vl_hidden"][i, :n], new_meta, source_chunk=name)
            kept += 1
        # one output chunk per source chunk, whatever the writer's chunk size
        writer.flush()
        stats["new"] += kept
        print(f"[extract-1f] {name}: kept {kept}/{len(chunk['metadata'])}", flush=True)
    return stats


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Re-extract drafter hidden states, reuse stored verifier states")
    p.add_argument("--existing_chunks", required=True)
    p.add_argument("--dataset_dir", required=True)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--split", default="train", choices=["train", "test"])
    p.add_argument("--event_image_key", default="event_image_1f")
    p.add_argument("--max_new_tokens", type=int, default=50)
    p.add_argument("--questions_file", default=None)
    p.add_argument("--max_questions", type=int, default=10)
    p.add_argument("--chunk_size", type=int, default=1000)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--egpt_model_path", default=None)
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"])
    p.add_argument("--single_adaptor", action="store_true")
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    return p


def main(argv=None) -> Dict[str, int]:
    from ..data.tokenizer import ByteTokenizer, load_tokenizer
    from ..models import EventChatConfig, EventChatModel
    args = build_parser().parse_args(argv)
    dtype = torch.bfloat16 if str(args.device).startswith("cuda") else torch.float32
    if args.random_init:
        cfg = EventChatConfig.eventgpt_7b() if args.random_init == "7b" else EventChatConfig.tiny()
        model, tok = EventChatModel.from_config(cfg, dtype, args.device, seed=0), ByteTokenizer()
    else:
        model = EventChatModel.from_pretrained(args.egpt_model_path, dtype, device=args.device)
        tok = load_tokenizer(args.egpt_model_path)
    ds = InstructionDataset(args.dataset_dir)
    questions = load_questions(args.questions_file or os.path.join(args.dataset_dir, "top50_questions.json"),
                               args.max_questions)
    src_cfg = read_index(args.existing_chunks).get("config", {})
    suffix = args.event_image_key.replace("event_image_", "")
    out = os.path.join(args.output_dir, f"chunked_{args.split}_{src_cfg.get('duration', '1s')}_"
                                        f"{src_cfg.get('quant', 'bf16')}_{suffix}")
    if not args.resume and os.path.exists(os.path.join(out, "index.json")):
        raise SystemExit(f"{out} already holds an index; pass --resume to continue it")
    writer = ChunkedHiddenStateWriter(out, args.chunk_size, dict(
        src_cfg, event_image_key=args.event_image_key, source_chunks=str(args.existing_chunks),
        max_new_tokens=args.max_new_tokens))
    ex = HiddenStateExtractor(model, tok, None, None, args.max_new_tokens, args.event_image_key,
                              double_adaptor=not args.single_adaptor)
    stop = StopFlag().install()
    try:
        stats = reextract(ex, ds, questions, args.existing_chunks, writer, stop)
    finally:
        stop.restore()
    print(f"[extract-1f] done: {stats} -> {out}", flush=True)
    return stats


if __name__ == "__main__":
    main()
