"""Parallel-prefill benchmark: drafter and verifier run their five stages on the
same samples; per-token decode times, cross-tokenizer acceptance, and the
overlap model "how many draft tokens hide inside the verifier's longer prefill".

Counterpart of feasible/benchmark_parallel_prefill/benchmark_parallel_prefill_5stages.py
(acceptance :216-293, the two per-model runs :296-630, overlap model :633-685,
statistics :785-872, report :875-1011, flags :1014-1030).  Differences that come
with the runtime:

  * per-token times are taken from HIP events recorded after every graph replay,
    so measuring them does not add a host sync per token;
  * next to the *modelled* overlap (from separately measured stage times, which is
    all the reference reports) the benchmark runs both prefills concurrently on
    two streams and reports the measured wall clock and the number of draft
    tokens that really were produced inside the window
    (``DualModelScheduler.hidden_prefill``).
"""
from __future__ import annotations

import argparse
import os
import time
from datetime import datetime
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..eval.reports import markdown_table, write_json
from ..eval.timers import StageTimer, gpu_memory_mb
from ..pipeline.dataset import DEFAULT_QUESTION, InstructionDataset
from .stages import (STAGE_KEYS, EventGPTStages, VideoLlavaStages, _use_engine,  # noqa: F401
                     compute_aligned_acceptance_rate, frames_to_grid)


# ------------------------------------------------------------------ acceptance
def retokenized_acceptance(draft_text: str, target_text: str, target_tokenizer=None) -> Dict:
    """Position-wise token agreement after both texts are tokenised by the *target*
    tokenizer (the two models do not share a vocabulary); the denominator is the
    target length.  Without a tokenizer: word-set overlap."""
    if target_tokenizer is None:
        a, b = set(draft_text.lower().split()), set(target_text.lower().split())
        total = max(len(a), len(b), 1)
        hit = len(a & b) if draft_text and target_text else 0
        return {"acceptance_rate": hit / total, "matched_tokens": hit, "total_tokens": total,
                "method": "fallback_text_overlap"}

    def ids(text):
        out = target_tokenizer(text)
        if isinstance(out, (list, tuple)):
            return list(out)
        return list(out["input_ids"])

    d, t = ids(draft_text), ids(target_text)
    if not d or not t:
        return {"acceptance_rate": 0.0, "matched_tokens": 0, "total_tokens": max(len(d), len(t)),
                "method": "token_level_mismatch"}
    hit = sum(1 for x, y in zip(d, t) if x == y)
    return {"acceptance_rate": hit / len(t), "matched_tokens": hit, "total_tokens": len(t),
            "draft_tokens": len(d), "method": "token_level_matching"}


# ---------------------------------------------------------------- overlap model
def compute_parallel_metrics(egpt: Dict, vl: Dict) -> Dict:
    """Both pipelines start together.  The window between the two prefill-complete
    times is free drafting time; decode that does not fit in it runs alongside the
    verifier's decode.  All times in seconds, as in the stage records."""
    e_done, v_done = egpt["prefill_complete_time"], vl["prefill_complete_time"]
    window = max(0.0, v_done - e_done)
    rate = egpt.get("decode_tokens_per_sec", 0.0)
    hidden = min(int(window * rate), egpt.get("decode_tokens", 0)) if rate > 0 else 0
    e_dec, v_dec = egpt["stage5_time"], vl["stage5_time"]
    hidden_time = min(window, e_dec)
    wall_prefill = max(e_done, v_done)
    wall_decode = max(e_dec - hidden_time, v_dec)
    wall = wall_prefill + wall_decode
    seq = egpt["total_time"] + vl["total_time"]
    return {"egpt_prefill_complete": e_done, "vl_prefill_complete": v_done, "overlap_window": window,
            "hidden_tokens": hidden, "hidden_decode_time": hidden_time, "wall_clock_prefill": wall_prefill,
            "wall_clock_decode": wall_decode, "wall_clock_total": wall, "sequential_total": seq,
            "parallel_speedup": seq / wall if wall > 0 else 0.0,
            "egpt_tokens_generated": egpt.get("decode_tokens", 0),
            "vl_tokens_generated": vl.get("decode_tokens", 0)}


# ------------------------------------------------------------- per-token times
@torch.no_grad()
def decode_token_times(model, embeds: torch.Tensor, max_new_tokens: int, eos: int) -> Dict:
    """Prefill + greedy decode with one timestamp per emitted token (ms since the
    start of the prefill).  Engine path: HIP events, one sync at the end."""
    lm = getattr(model, "language_model", model)
    if _use_engine(model, embeds):
        eng = model.engine()
        eng.eos = eos
        n_max = min(max_new_tokens, eng.max_out)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_max + 1)]
        torch.cuda.synchronize(embeds.device)
        evs[0].record()
        eng.reset()
        eng.prefill(embeds, select="greedy")
        evs[1].record()
        for i in range(1, n_max):
            eng.decode_step(True)
            evs[i + 1].record()
        torch.cuda.synchronize(embeds.device)
        n = min(int(eng.step[0].item()), n_max)       # the step counter stops at EOS
        stamps = [evs[0].elapsed_time(evs[i + 1]) for i in range(n)]
        toks = eng.out_tokens[0, :n].tolist()
    else:
        t0 = time.perf_counter()
        out = lm.model(inputs_embeds=embeds[None], use_cache=True)
        nxt = lm.lm_head(out.last_hidden_state[:, -1]).float().argmax(-1)
        toks, stamps, past, pos = [int(nxt)], [(time.perf_counter() - t0) * 1e3], out.past_key_values, embeds.shape[0]
        while len(toks) < max_new_tokens and toks[-1] != eos:
            out = lm.model(inputs_embeds=lm.model.embed_tokens(nxt)[:, None],
                           position_ids=torch.full((1, 1), pos, device=embeds.device),
                           past_key_values=past, use_cache=True)
            past, pos = out.past_key_values, pos + 1
            nxt = lm.lm_head(out.last_hidden_state[:, -1]).float().argmax(-1)
            toks.append(int(nxt))
            stamps.append((time.perf_counter() - t0) * 1e3)
    gaps = np.diff(stamps) if len(stamps) > 1 else np.zeros(0)
    return {"token_ids": toks, "token_times_ms": [float(s) for s in stamps], "prefill_ms": float(stamps[0]),
            "per_token_ms_mean": float(gaps.mean()) if gaps.size else 0.0,
            "per_token_ms_p50": float(np.median(gaps)) if gaps.size else 0.0,
            "per_token_ms_max": float(gaps.max()) if gaps.size else 0.0}


# ------------------------------------------------------------------ the runner
def _with_prefill_complete(rec: Dict) -> Dict:
    rec["prefill_complete_time"] = sum(rec[k] for k in STAGE_KEYS[:4])
    rec["tokens_per_sec"] = rec["decode_tokens_per_sec"]
    rec["num_output_tokens"] = rec["decode_tokens"]
    return rec


class ParallelPrefillBenchmark:
    def __init__(self, egpt, egpt_tok, vl, vl_proc, max_new_tokens: int = 50, num_frames: int = 1,
                 event_image_key: str = "event_image_1f", token_adapter=None, live: Optional[bool] = None):
        self.egpt, self.vl = egpt, vl
        self.e_run = EventGPTStages(egpt, egpt_tok, num_frames, "image", event_image_key, max_new_tokens)
        self.v_run = VideoLlavaStages(vl, vl_proc, max_new_tokens)
        self.vl_tok = getattr(vl_proc, "tokenizer", None)
        self.adapter = token_adapter
        self.max_new = max_new_tokens
        on_gpu = egpt.device.type == "cuda" and egpt.dtype == torch.bfloat16
        self.live = on_gpu if live is None else (live and on_gpu)
        self._sched = None

    def _scheduler(self):
        if self._sched is None:
            from ..runtime.scheduler import DualModelScheduler
            self._sched = DualModelScheduler(self.egpt.engine(), self.vl.engine())
        return self._sched

    def run_sample(self, ds: InstructionDataset, sample: Dict, query: str) -> Optional[Dict]:
        e = self.e_run.run(ds, sample, query)
        v = self.v_run.run(ds, sample, query)
        if e is None or v is None:
            return None
        e, v = _with_prefill_complete(e), _with_prefill_complete(v)
        rec = {"egpt": e, "vl": v, "parallel": compute_parallel_metrics(e, v),
               "acceptance": retokenized_acceptance(e["output"], v["output"], self.vl_tok)}
        if self.adapter is not None:
            rec["aligned_acceptance"] = compute_aligned_acceptance_rate(
                e["token_ids"], v["token_ids"], self.adapter, str(self.egpt.device))
        return rec

    @torch.no_grad()
    def run_live(self, d_embeds: torch.Tensor, v_embeds: torch.Tensor, eos: int = 2) -> Dict:
        """Measured (not modelled) hiding: both prefills at once, drafter free-runs."""
        s = self._scheduler()
        seq = s.sequential_prefill(d_embeds, v_embeds)
        par = s.parallel_prefill(d_embeds, v_embeds)
        hid = s.hidden_prefill(d_embeds, v_embeds, max_free=self.max_new, eos_token_id=eos)
        return {"sequential_prefill_ms": seq["sequential_ms"], "parallel_prefill_wall_ms": par["wall_ms"],
                "prefill_overlap_ratio": par["overlap_ratio"], "hidden_tokens_measured": hid["n_free"],
                "hidden_window_wall_ms": hid["prefill_wall_ms"], "egpt_prefill_ms": hid["egpt_prefill_ms"],
                "vl_prefill_ms": hid["vl_prefill_ms"]}


def compute_statistics(results: List[Dict]) -> Dict:
    def stat(vals):
        vals = [float(x) for x in vals]
        return {"avg": float(np.mean(vals)) if vals else 0.0, "std": float(np.std(vals)) if vals else 0.0,
                "min": float(np.min(vals)) if vals else 0.0, "max": float(np.max(vals)) if vals else 0.0}

    out: Dict = {"n_samples": len(results)}
    for side in ("egpt", "vl"):
        for k in STAGE_KEYS + ("total_time", "prefill_complete_time", "decode_tokens_per_sec", "decode_tokens",
                               "prefill_tokens", "kv_cache_mb"):
            out[f"{side}_{k}"] = stat(r[side][k] for r in results)
    for k in ("overlap_window", "hidden_tokens", "wall_clock_total", "sequential_total", "parallel_speedup"):
        out[k] = stat(r["parallel"][k] for r in results)
    out["acceptance_rate"] = stat(r["acceptance"]["acceptance_rate"] for r in results)
    if results and "aligned_acceptance" in results[0]:
        out["aligned_acceptance_rate"] = stat(r["aligned_acceptance"]["aligned_acceptance_rate"] for r in results)
    live = [r["live"] for r in results if "live" in r]
    if live:
        for k in live[0]:
            out[f"live_{k}"] = stat(x[k] for x in live)
    return out


def write_report(stats: Dict, results: List[Dict], out_dir: str, ts: str, config: Dict) -> Dict[str, str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = {"json": write_json({"timestamp": ts, "config": config, "statistics": stats, "results": results},
                                os.path.join(out_dir, f"parallel_prefill_5stages_{ts}.json"))}
    ms = lambda s: f"{s['avg'] * 1e3:.2f} ± {s['std'] * 1e3:.2f}"                     # noqa: E731
    rows = [[name, ms(stats[f"egpt_{k}"]), ms(stats[f"vl_{k}"])]
            for name, k in zip(("S1 load", "S2 preprocess", "S3 vision", "S4 prefill", "S5 decode"), STAGE_KEYS)]
    rows.append(["prefill complete", ms(stats["egpt_prefill_complete_time"]), ms(stats["vl_prefill_complete_time"])])
    rows.append(["total", ms(stats["egpt_total_time"]), ms(stats["vl_total_time"])])
    md = [f"# Parallel prefill benchmark ({ts})", "",
          f"data: {config['data']}, weights: {config['weights']}, samples: {stats['n_samples']}", "",
          markdown_table(["stage (ms)", "drafter", "verifier"], rows), "",
          "## Overlap (modelled from the stage times)", "",
          markdown_table(["metric", "value"], [
              ("overlap window ms", f"{stats['overlap_window']['avg'] * 1e3:.2f}"),
              ("hidden draft tokens", f"{stats['hidden_tokens']['avg']:.1f}"),
              ("parallel wall clock ms", f"{stats['wall_clock_total']['avg'] * 1e3:.2f}"),
              ("sequential ms", f"{stats['sequential_total']['avg'] * 1e3:.2f}"),
              ("parallel speedup", f"{stats['parallel_speedup']['avg']:.2f}x"),
              ("token acceptance (re-tokenised)", f"{100 * stats['acceptance_rate']['avg']:.2f} %")]
              + ([("token acceptance (token adapter)", f"{100 * stats['aligned_acceptance_rate']['avg']:.2f} %")]
                 if "aligned_acceptance_rate" in stats else [])), ""]
    live = [(k[5:], f"{v['avg']:.2f}") for k, v in stats.items() if k.startswith("live_")]
    if live:
        md += ["## Overlap (measured: both models on one GPU, two streams)", "",
               markdown_table(["metric", "value"], live), ""]
    paths["md"] = os.path.join(out_dir, f"parallel_prefill_5stages_{ts}.md")
    with open(paths["md"], "w") as f:
        f.write("\n".join(md))
    return paths


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="parallel-prefill 5-stage benchmark, drafter vs verifier")
    p.add_argument("--dataset_dir", default=None, help="instruction dataset; synthetic if omitted")
    p.add_argument("--eventgpt_model_path", default=None)
    p.add_argument("--videollava_model", default=None)
    from ..quant import add_quant_args
    add_quant_args(p)
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"])
    p.add_argument("--max_samples", type=int, default=-1)
    p.add_argument("--max_new_tokens", type=int, default=50)
    p.add_argument("--output_dir", default=None)
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--num_frames", type=int, default=1)
    p.add_argument("--use_token_adapter", action="store_true")
    p.add_argument("--token_adapter_path", default=None)
    p.add_argument("--no_live", action="store_true", help="skip the measured two-stream overlap")
    p.add_argument("--warmup", type=int, default=1)
    return p


def main(argv=None) -> Dict:
    from ..pipeline.extract_hidden_states import load_models
    from .e2e_wallclock import SampleInputs
    args = build_parser().parse_args(argv)
    ts = datetime.now().strftime("%Y%m%d_%H%M%S")
    data_kind = "real"
    if args.dataset_dir is None:
        import tempfile

        from ..pipeline.dataset import make_synthetic_dataset
        big = args.random_init != "tiny"
        n = (args.max_samples if args.max_samples > 0 else 4) + args.warmup
        args.dataset_dir = make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_pp_"), n_samples=n,
                                                  n_events=130000 if big else 4000, width=640 if big else 160,
                                                  height=480 if big else 120)
        data_kind = "synthetic"
    ds = InstructionDataset(args.dataset_dir, args.max_samples + args.warmup if args.max_samples > 0 else -1)
    largs = argparse.Namespace(device=args.device, random_init=args.random_init,
                               egpt_model_path=args.eventgpt_model_path, vl_model_path=args.videollava_model,
                               use_4bit=args.use_4bit, quant_type=args.quant_type)
    egpt, tok, vl, vproc = load_models(largs)
    adapter = None
    if args.use_token_adapter and args.token_adapter_path:
        from ..align.token_align import TokenAdapter
        adapter = TokenAdapter.load(args.token_adapter_path, args.device)
    key = "event_image_1f" if args.num_frames == 1 else "event_image"
    bench = ParallelPrefillBenchmark(egpt, tok, vl, vproc, args.max_new_tokens, args.num_frames, key, adapter,
                                     live=not args.no_live)
    prep = SampleInputs(egpt, tok, vl, vproc, key)
    results: List[Dict] = []
    for i, sample in enumerate(ds.samples):
        q = ds.question(sample) or DEFAULT_QUESTION
        rec = bench.run_sample(ds, sample, q)
        if rec is None or i < args.warmup:
            continue
        inp = prep.load(ds, sample, q)
        if inp is not None:
            d_emb, v_emb = prep.egpt_embeds(inp), prep.vl_embeds(inp)
            eos = egpt.config.eos_token_id if egpt.config.eos_token_id is not None else 2
            rec["egpt"]["token_times"] = decode_token_times(egpt, d_emb, args.max_new_tokens, eos)
            rec["vl"]["token_times"] = decode_token_times(vl, v_emb, args.max_new_tokens, eos)
            if bench.live:
                rec["live"] = bench.run_live(d_emb, v_emb, eos)
        rec["sample"] = i
        results.append(rec)
    stats = compute_statistics(results)
    config = {"data": data_kind, "weights": args.random_init or "checkpoint", "weight_format": args.quant_type if args.use_4bit else "bf16", "num_frames": args.num_frames,
              "max_new_tokens": args.max_new_tokens, "device": str(args.device), "memory": gpu_memory_mb()}
    out_dir = args.output_dir or os.path.join(os.getcwd(), "tasks")
    paths = write_report(stats, results, out_dir, ts, config)
    print(f"[parallel_prefill] {stats['n_samples']} samples; overlap window "
          f"{stats['overlap_window']['avg'] * 1e3:.1f} ms, hidden tokens {stats['hidden_tokens']['avg']:.1f}, "
          f"parallel speedup {stats['parallel_speedup']['avg']:.2f}x; wrote {paths['md']}")
    return {"statistics": stats, "results": results, "paths": paths}


if __name__ == "__main__":
    main()
