"""Prefill-hiding benchmark runner: the three runtimes of ``runtime.prefill_hiding``
against the verifier's own greedy decode, over scenes x questions, plus the
closed-form speedup calculator.

Counterpart of feasible/egpt_prefill_only/benchmark_prefill_only.py:369-378 (flags,
per-sample loop, report), run_full_benchmark.py:313-316 (scenes x top-N questions)
and compute_speedup.py:234-237 (``--acceptance_rate`` / ``--measure``).  Everything
measured here is live: the drafter really decodes while the verifier prefills, and
every speculative output is compared token-for-token with the baseline.

    python -m eventgpt_amd.benchmarks.prefill_hiding_bench --random_init small --max_samples 4
    python -m eventgpt_amd.benchmarks.prefill_hiding_bench --theory --egpt_prefill_ms 13 \
           --vl_prefill_ms 41 --vl_token_ms 3.3 --egpt_token_ms 2.9 --verify_ms 4.4 --acceptance_rate 0.279
"""
from __future__ import annotations

import argparse
import os
import time
from datetime import datetime
from typing import Dict, List, Optional

import numpy as np
import torch

from ..eval.reports import markdown_table, write_json
from ..runtime.spec_decode import matches_greedy
from ..runtime.prefill_hiding import (AlignedPrefillInference, PrefillOnlyInference, PrefillThenVerifyInference,
                                      compute_theoretical_speedup, evaluate_sd_outputs)

METHODS = ("prefill_only", "prefill_then_verify", "aligned")


def theory_table(egpt_prefill_ms: float, vl_prefill_ms: float, vl_token_ms: float, egpt_token_ms: float,
                 verify_ms: float, max_new_tokens: int = 50,
                 acceptance_rates=(0.05, 0.10, 0.20, 0.279, 0.40, 0.60, 0.80, 1.0)) -> List[Dict]:
    return [dict(acceptance_rate=a, **compute_theoretical_speedup(egpt_prefill_ms, vl_prefill_ms, vl_token_ms,
                                                                  max_new_tokens, a, egpt_token_ms, verify_ms))
            for a in acceptance_rates]


def _mean(xs):
    xs = list(xs)
    return float(np.mean(xs)) if xs else 0.0


class PrefillHidingBenchmark:
    def __init__(self, egpt, etok, vl, vproc, methods=METHODS, adapter: Optional[torch.nn.Module] = None,
                 token_map=None, max_new_tokens: int = 50, max_draft_tokens: int = 30,
                 event_image_key: str = "event_image"):
        from .e2e_wallclock import SampleInputs
        self.prep = SampleInputs(egpt, etok, vl, vproc, event_image_key)
        d, v = egpt.engine(), vl.engine()
        vtok = getattr(vproc, "tokenizer", None)
        self.eos = vl.config.text_config.eos_token_id if vl.config.text_config.eos_token_id is not None else 2
        self.max_new = max_new_tokens
        self.runners: Dict[str, object] = {}
        if "prefill_only" in methods:
            self.runners["prefill_only"] = PrefillOnlyInference(d, v, etok, vtok, max_new_tokens)
        if "prefill_then_verify" in methods:
            self.runners["prefill_then_verify"] = PrefillThenVerifyInference(
                d, v, token_map, vtok, max_new_tokens, max_draft_tokens)
        if "aligned" in methods and adapter is not None:
            self.runners["aligned"] = AlignedPrefillInference(
                d, v, adapter, vl.language_model.lm_head.weight.data, verifier_tokenizer=vtok,
                max_new_tokens=max_new_tokens, max_draft_tokens=max_draft_tokens)
        self.sched = next(iter(self.runners.values())).sched

    @torch.no_grad()
    def run_sample(self, inp: Dict) -> Dict:
        d_emb, v_emb = self.prep.egpt_embeds(inp), self.prep.vl_embeds(inp)
        base = self.sched.verifier_baseline(v_emb, self.max_new, self.eos)
        base_rec = {"tokens": base["tokens"].tolist(), "prefill_ms": base["prefill_ms"],
                    "decode_ms": base["decode_ms"], "total_ms": base["prefill_ms"] + base["decode_ms"]}
        rec: Dict = {"baseline": base_rec}
        for name, r in self.runners.items():
            out = r.generate(d_emb, v_emb, self.eos).as_dict()
            # token-identical, or first differing at a near-tie of the verifier's own logits
            chk = matches_greedy(self.sched.v, v_emb,
                                 torch.tensor(out["output_tokens"], dtype=torch.long),
                                 base["tokens"].long().cpu())
            out["identical_to_baseline"] = bool(chk["lossless"])
            out["exact_match"] = bool(chk["exact"])
            out["mismatch_logit_gap"] = chk["mismatch_logit_gap"]
            rec[name] = out
        return rec


def aggregate(samples: List[Dict]) -> Dict:
    agg: Dict = {"n": len(samples)}
    if not samples:
        return agg
    base = [s["baseline"] for s in samples]
    agg["baseline"] = {"prefill_ms": _mean(b["prefill_ms"] for b in base), "decode_ms": _mean(b["decode_ms"] for b in base),
                       "total_ms": _mean(b["total_ms"] for b in base),
                       "tokens": _mean(len(b["tokens"]) for b in base)}
    for m in METHODS:
        rows = [s[m] for s in samples if m in s]
        if not rows:
            continue
        a = {"wall_ms": _mean(r["wall_clock_time"] * 1e3 for r in rows),
             "identical_outputs": sum(bool(r["identical_to_baseline"]) for r in rows),
             "egpt_prefill_ms": _mean(r["egpt_prefill_time"] * 1e3 for r in rows),
             "vl_prefill_ms": _mean(r["vl_prefill_time"] * 1e3 for r in rows)}
        a["speedup_vs_baseline"] = agg["baseline"]["total_ms"] / a["wall_ms"] if a["wall_ms"] > 0 else 1.0
        if m == "prefill_only":
            a["free_tokens"] = _mean(r["free_tokens_count"] for r in rows)
            a["overlap_window_ms"] = _mean(r["overlap_window"] * 1e3 for r in rows)
            a["speedup_vs_sequential"] = _mean(r["total_speedup_vs_sequential"] for r in rows)
        else:
            a.update(evaluate_sd_outputs(rows, base[: len(rows)]))
            a["drafted"] = _mean(r["n_drafted"] for r in rows)
        agg[m] = a
    return agg


def write_report(result: Dict, out_dir: str, ts: str) -> Dict[str, str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = {"json": write_json(result, os.path.join(out_dir, f"prefill_hiding_{ts}.json"))}
    agg = result["aggregate"]
    md = [f"# Prefill-hiding benchmark ({ts})", "",
          f"data: {result['config']['data']}, weights: {result['config']['weights']}, samples: {agg['n']}", ""]
    if agg.get("baseline"):
        b = agg["baseline"]
        rows = [["verifier baseline", f"{b['total_ms']:.1f}", "1.00x", "-", "-", "-"]]
        for m in METHODS:
            if m in agg:
                a = agg[m]
                rows.append([m, f"{a['wall_ms']:.1f}", f"{a['speedup_vs_baseline']:.2f}x",
                             f"{a.get('drafted', a.get('free_tokens', 0)):.1f}",
                             f"{100 * a['accept_rate']:.1f} %" if "accept_rate" in a else "-",
                             f"{a['identical_outputs']}/{agg['n']}"])
        md += [markdown_table(["method", "wall ms", "speedup", "free/drafted tokens", "accepted", "identical outputs"],
                              rows), ""]
    if result.get("theory"):
        md += ["## Closed-form estimate", "",
               markdown_table(["acceptance", "free drafts", "accepted", "saved ms", "SD ms", "speedup"],
                              [[f"{t['acceptance_rate']:.3f}", t["free_drafts"], t["accepted"],
                                f"{t['time_saved_ms']:.1f}", f"{t['sd_time_ms']:.1f}", f"{t['e2e_speedup']:.3f}x"]
                               for t in result["theory"]]), ""]
    paths["md"] = os.path.join(out_dir, f"prefill_hiding_{ts}.md")
    with open(paths["md"], "w") as f:
        f.write("\n".join(md))
    return paths


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="prefill-hiding runtimes vs verifier baseline")
    p.add_argument("--dataset_dir", default=None)
    p.add_argument("--max_samples", type=int, default=10)
    p.add_argument("--num_questions", type=int, default=1, help="top N questions per scene")
    p.add_argument("--questions_file", default=None)
    p.add_argument("--max_new_tokens", "--max_tokens", dest="max_new_tokens", type=int, default=50)
    p.add_argument("--max_draft_tokens", type=int, default=30)
    p.add_argument("--output_dir", default=None)
    p.add_argument("--methods", default=",".join(METHODS))
    p.add_argument("--use_alignment", action="store_true", help="map draft ids with a trained TokenAdapter")
    p.add_argument("--adapter_path", default=None, help="TokenAdapter (--use_alignment) checkpoint")
    p.add_argument("--hidden_adapter", default=None, help="hidden-state adapter checkpoint for the aligned method")
    p.add_argument("--adapter_dir", default=None)
    p.add_argument("--egpt_model_path", default=None)
    p.add_argument("--vl_model_path", default=None)
    from ..quant import add_quant_args
    add_quant_args(p)
    p.add_argument("--random_init", default=None, choices=["7b", "small"])
    p.add_argument("--event_image_key", default="event_image")
    p.add_argument("--warmup", type=int, default=1)
    # closed-form mode
    p.add_argument("--theory", action="store_true", help="only print the closed-form table")
    p.add_argument("--measure", action="store_true", help="(default unless --theory) run the measurements")
    p.add_argument("--acceptance_rate", type=float, default=0.279)
    p.add_argument("--egpt_prefill_ms", type=float, default=None)
    p.add_argument("--vl_prefill_ms", type=float, default=None)
    p.add_argument("--vl_token_ms", type=float, default=None)
    p.add_argument("--egpt_token_ms", type=float, default=None)
    p.add_argument("--verify_ms", type=float, default=None)
    return p


def main(argv=None) -> Dict:
    args = build_parser().parse_args(argv)
    ts = datetime.now().strftime("%Y%m%d_%H%M%S")
    out_dir = args.output_dir or os.path.join(os.getcwd(), "tasks")
    rates = sorted({0.05, 0.10, 0.20, 0.40, 0.60, 0.80, 1.0, args.acceptance_rate})
    if args.theory:
        need = ("egpt_prefill_ms", "vl_prefill_ms", "vl_token_ms", "egpt_token_ms", "verify_ms")
        miss = [n for n in need if getattr(args, n) is None]
        if miss:
            raise SystemExit(f"--theory needs {', '.join('--' + m for m in miss)}")
        result = {"timestamp": ts, "config": {"data": "none", "weights": "none"}, "aggregate": {"n": 0},
                  "theory": theory_table(args.egpt_prefill_ms, args.vl_prefill_ms, args.vl_token_ms,
                                         args.egpt_token_ms, args.verify_ms, args.max_new_tokens, rates)}
        result["paths"] = write_report(result, out_dir, ts)
        return result
    if not torch.cuda.is_available():
        raise SystemExit("the measured mode runs on the serving engines and needs a GPU (use --theory on CPU)")
    from ..pipeline.dataset import DEFAULT_QUESTION, InstructionDataset, load_questions
    from .e2e_wallclock import load_everything
    data_kind = "real"
    if args.dataset_dir is None:
        import tempfile

        from ..pipeline.dataset import make_synthetic_dataset
        args.dataset_dir = make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_ph_"),
                                                  n_samples=args.max_samples + args.warmup, n_events=60000,
                                                  width=640, height=480)
        data_kind = "synthetic"
    ds = InstructionDataset(args.dataset_dir, args.max_samples + args.warmup)
    qfile = args.questions_file or os.path.join(args.dataset_dir, "top50_questions.json")
    questions = load_questions(qfile, args.num_questions) if os.path.exists(qfile) else [DEFAULT_QUESTION]
    egpt, etok, vl, vproc, adapters = load_everything(args)
    adapter = None
    if args.hidden_adapter:
        from ..align.adapters import load_any_adapter
        adapter = load_any_adapter(args.hidden_adapter, "cuda")[0]
    elif adapters:
        adapter = next((adapters[k] for k in ("L4", "L3", "L2", "L1") if k in adapters), None)
    token_map = None
    if args.use_alignment and args.adapter_path:
        from ..align.token_align import TokenAdapter, TokenAdapterForSD
        token_map = TokenAdapterForSD(TokenAdapter.load(args.adapter_path, "cuda"))
    bench = PrefillHidingBenchmark(egpt, etok, vl, vproc, args.methods.split(","), adapter, token_map,
                                   args.max_new_tokens, args.max_draft_tokens, args.event_image_key)
    samples: List[Dict] = []
    for si, sample in enumerate(ds.samples):
        for qi, q in enumerate(questions):
            inp = bench.prep.load(ds, sample, q)
            if inp is None:
                continue
            r = bench.run_sample(inp)
            r.update(sample=si, question=qi)
            if si >= args.warmup:
                samples.append(r)
    agg = aggregate(samples)
    result = {"timestamp": ts, "config": {"data": data_kind, "weights": args.random_init or "checkpoint", "weight_format": args.quant_type if args.use_4bit else "bf16",
                                          "max_new_tokens": args.max_new_tokens,
                                          "max_draft_tokens": args.max_draft_tokens, "methods": list(bench.runners)},
              "aggregate": agg, "samples": samples}
    if agg.get("baseline") and "prefill_then_verify" in agg:
        b, a = agg["baseline"], agg["prefill_then_verify"]
        tok_ms = b["decode_ms"] / max(b["tokens"] - 1, 1)
        po = agg.get("prefill_only", {})
        e_tok = (po["overlap_window_ms"] / po["free_tokens"]) if po.get("free_tokens") else tok_ms
        result["theory"] = theory_table(a["egpt_prefill_ms"], a["vl_prefill_ms"], tok_ms, e_tok, tok_ms,
                                        args.max_new_tokens, rates)
    result["paths"] = write_report(result, out_dir, ts)
    for m in bench.runners:
        a = agg.get(m, {})
        if a:
            print(f"[prefill_hiding] {m:20s} wall {a['wall_ms']:8.1f} ms  {a['speedup_vs_baseline']:.2f}x  "
                  f"identical {a['identical_outputs']}/{agg['n']}", flush=True)
    return result


if __name__ == "__main__":
    main()
