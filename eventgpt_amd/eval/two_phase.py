"""Two-phase pipeline evaluation: prefill hiding (L1-L4) + decode drafting (L5F / B1).

Phase 1 compares ``adapter(h_egpt)[t]`` against the verifier's token at the same
position inside the free-draft window; phase 2 compares the decode adapter's
prediction at ``t`` against the verifier's token at ``t+1``; both feed an analytic
end-to-end estimate (reference: pipeline/evaluation/eval_two_phase.py:36-308, same
flags and result keys). The two phases are pure functions of token-id matrices so
they can be tested without adapters or LM heads.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, Optional

import torch

from .acceptance import (NUM_OUTPUT_TOKENS, TimingConfig, apply_adapter, consecutive_run, load_eval_data,
                         load_lm_head, project_to_tokens)
from .reports import write_json


def phase1_prefill(prefill_tokens: torch.Tensor, vl_tokens: torch.Tensor, mask: torch.Tensor,
                   gamma_prefill: int, timing: TimingConfig) -> Dict[str, float]:
    mb = mask.bool()
    match = (prefill_tokens == vl_tokens) & mb
    gp = min(gamma_prefill, match.shape[1])
    free = consecutive_run(match[:, :gp]).mean().item() if gp > 0 else 0.0
    return {"phase1_prefill_top1": float(match.sum()) / max(float(mb.sum()), 1.0),
            "phase1_free_tokens": free, "phase1_time_saved_ms": free * timing.vl_per_token}


def phase2_decode(decode_tokens: torch.Tensor, decode_topk: torch.Tensor, vl_tokens: torch.Tensor,
                  mask: torch.Tensor, gamma_decode: int) -> Dict:
    mb = mask.bool()
    n_valid = max(float(mb.sum()), 1.0)
    out: Dict = {"phase2_same_pos_top1": float(((decode_tokens == vl_tokens) & mb).sum()) / n_valid}
    S = decode_tokens.shape[1]
    if S <= 1:
        out.update(phase2_shifted_top1=0.0, phase2_shifted_top5=0.0, phase2_decode_consecutive=0.0,
                   phase2_shifted_consecutive=0.0, phase2_position_shifted_top1=[])
        return out
    target, smask = vl_tokens[:, 1:], mb[:, 1:]
    match = (decode_tokens[:, :-1] == target) & smask
    n_s = max(float(smask.sum()), 1.0)
    top5 = (decode_topk[:, :-1, :5] == target.unsqueeze(-1)).any(-1) & smask
    run_all = consecutive_run(match).mean().item()
    run_g = consecutive_run(match[:, :gamma_decode]).mean().item() if S - 1 >= gamma_decode else run_all
    per_pos = []
    for pos in range(min(10, S - 1)):
        col = smask[:, pos]
        if not col.any():
            break
        per_pos.append(match[col, pos].float().mean().item())
    out.update(phase2_shifted_top1=float(match.sum()) / n_s, phase2_shifted_top5=float(top5.sum()) / n_s,
               phase2_decode_consecutive=run_g, phase2_shifted_consecutive=run_all,
               phase2_position_shifted_top1=per_pos)
    return out


def combined_speedup(free_tokens: float, decode_run: float, timing: TimingConfig,
                     n_tokens: int = NUM_OUTPUT_TOKENS) -> Dict[str, float]:
    """One batched verifier pass per SD iteration (eval_two_phase.py:239-265)."""
    remaining = max(0.0, n_tokens - free_tokens)
    if decode_run > 0:
        iters = remaining / (decode_run + 1)
        decode_ms = iters * (timing.vl_per_token + timing.adapter_latency)
    else:
        iters, decode_ms = remaining, remaining * timing.vl_per_token
    total = timing.parallel_prefill + decode_ms
    base = timing.baseline_ms(n_tokens)
    return {"baseline_time_ms": base, "sd_time_ms": total, "sd_iterations": iters,
            "e2e_speedup": base / total if total > 0 else 1.0}


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Two-phase (prefill hiding + decode) adapter evaluation")
    p.add_argument("--prefill_checkpoint", default=None)
    p.add_argument("--decode_checkpoint", required=True)
    p.add_argument("--test_data", required=True)
    p.add_argument("--lm_head", required=True)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--decode_vlm_only", action="store_true")
    p.add_argument("--gamma_decode", type=int, default=5)
    p.add_argument("--no_prefill", action="store_true")
    p.add_argument("--label", default=None)
    p.add_argument("--output_dir", default=None)
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    return p


def run(args, timing: Optional[TimingConfig] = None) -> Dict:
    from ..align.adapters import load_any_adapter
    t = timing or TimingConfig()
    gamma_prefill = int(t.free_draft_tokens)
    e, v, mask = load_eval_data(args.test_data)
    W = load_lm_head(args.lm_head)
    vl_tokens, _ = project_to_tokens(v, W, mask, 1, args.batch_size, args.device)

    results: Dict = {"label": args.label or ("decode_only" if args.no_prefill else "two_phase"),
                     "no_prefill": args.no_prefill, "prefill_adapter": args.prefill_checkpoint,
                     "decode_adapter": args.decode_checkpoint, "decode_vlm_only": args.decode_vlm_only,
                     "gamma_prefill": gamma_prefill, "gamma_decode": args.gamma_decode,
                     "num_output_tokens": NUM_OUTPUT_TOKENS}
    if args.no_prefill:
        results.update(phase1_prefill_top1=0.0, phase1_free_tokens=0.0, phase1_time_saved_ms=0.0)
    else:
        adapter, _ = load_any_adapter(args.prefill_checkpoint, args.device)
        aligned = apply_adapter(adapter.to(args.device).eval(), e, v, args.batch_size, args.device)
        p_tok, _ = project_to_tokens(aligned, W, mask, 1, args.batch_size, args.device)
        results.update(phase1_prefill(p_tok, vl_tokens, mask, gamma_prefill, t))
        del adapter, aligned

    adapter, _ = load_any_adapter(args.decode_checkpoint, args.device)
    aligned = apply_adapter(adapter.to(args.device).eval(), e, v, args.batch_size, args.device,
                            vlm_only=args.decode_vlm_only)
    d_tok, d_topk = project_to_tokens(aligned, W, mask, 10, args.batch_size, args.device)
    results["decode_adapter_class"] = type(adapter).__name__
    results.update(phase2_decode(d_tok, d_topk, vl_tokens, mask, args.gamma_decode))
    results.update(combined_speedup(results["phase1_free_tokens"], results["phase2_decode_consecutive"], t))
    return results


def output_path(args) -> str:
    out_dir = args.output_dir or os.path.join(os.getcwd(), "tasks")
    if args.label:
        name = f"two_phase_results_{args.label}.json"
    elif args.no_prefill:
        name = "decode_only_results.json"
    else:
        name = "two_phase_results.json"
    return os.path.join(out_dir, name)


def main(argv=None) -> Dict:
    parser = build_parser()
    args = parser.parse_args(argv)
    if not args.no_prefill and args.prefill_checkpoint is None:
        parser.error("--prefill_checkpoint is required unless --no_prefill is set")
    results = run(args)
    print(f"phase 1: top-1 {results['phase1_prefill_top1']:.2%}, free tokens "
          f"{results['phase1_free_tokens']:.2f}/{results['gamma_prefill']}")
    print(f"phase 2: shifted top-1 {results['phase2_shifted_top1']:.2%}, top-5 "
          f"{results['phase2_shifted_top5']:.2%}, run {results['phase2_decode_consecutive']:.2f}/"
          f"{results['gamma_decode']}")
    print(f"e2e: {results['baseline_time_ms']:.0f} ms -> {results['sd_time_ms']:.0f} ms "
          f"({results['e2e_speedup']:.2f}x)")
    print("saved", write_json(results, output_path(args)))
    return results


if __name__ == "__main__":
    main()
