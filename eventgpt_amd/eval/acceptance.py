"""Feature- and token-level speculative-decoding acceptance metrics.

Same metric keys and formulas as the reference's evaluator
(pipeline/evaluation/measure_feature_acceptance.py:43-917) so reports from either
side can be diffed; the structure is different:

* every metric is a pure function of a ``[N, S]`` cosine / match matrix and the
  validity mask; the only O(N*S*H) and O(N*S*V) passes (cosine, LM-head
  projection) are chunked and run where the data is (HIP GEMM with fp32 logits
  on a GPU, ``torch.matmul`` on the host),
* consecutive-accept lengths come from one ``cumprod`` over the sequence axis.

The timing model (``TimingConfig``) is a parameter everywhere; defaults are the
reference's RTX-4090 figures so analytic speedups are comparable, and
``TimingConfig.measured`` builds one from numbers measured on this machine.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass
from datetime import datetime
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

NUM_OUTPUT_TOKENS = 50          # length of the generation the analytic E2E model prices
CORRELATION_THRESHOLDS = (0.80, 0.85, 0.90, 0.92, 0.95, 0.98)


@dataclass
class TimingConfig:
    """Per-stage latencies in ms (measure_feature_acceptance.py:43-52)."""
    egpt_prefill: float = 130.0
    egpt_per_token: float = 25.0
    vl_prefill: float = 310.0
    vl_per_token: float = 25.0
    adapter_latency: float = 1.5

    @classmethod
    def measured(cls, drafter_prefill_ms: float, drafter_decode_ms: float, verifier_prefill_ms: float,
                 verifier_decode_ms: float, adapter_ms: float) -> "TimingConfig":
        return cls(drafter_prefill_ms, drafter_decode_ms, verifier_prefill_ms, verifier_decode_ms, adapter_ms)

    @property
    def parallel_prefill(self) -> float:
        return max(self.vl_prefill, self.egpt_prefill + self.adapter_latency)

    @property
    def free_draft_tokens(self) -> float:
        if self.egpt_per_token <= 0:
            return 0.0
        return max(0.0, self.vl_prefill - self.egpt_prefill) / self.egpt_per_token

    def baseline_ms(self, n_tokens: int = NUM_OUTPUT_TOKENS) -> float:
        return self.vl_prefill + n_tokens * self.vl_per_token


@dataclass
class SDConfig:
    gamma_decode: int = 5
    thresholds: Tuple[float, ...] = (0.80, 0.85, 0.90, 0.95)


# ------------------------------------------------------------------- primitives
def seq_mask(seq_lens: torch.Tensor, max_seq: int) -> torch.Tensor:
    return (torch.arange(max_seq, device=seq_lens.device)[None, :] < seq_lens[:, None]).float()


def masked_cosine(a: torch.Tensor, b: torch.Tensor, chunk: int = 256) -> torch.Tensor:
    """Row-wise cosine of two ``[N, S, H]`` tensors, fp32 accumulation, chunked over N."""
    if a.is_cuda:        # one fused pass over both tensors
        from .. import ops
        return ops.cosine_rows(a, b, 1e-12)
    out = torch.empty(a.shape[:2], dtype=torch.float32, device=a.device)
    for i in range(0, a.shape[0], chunk):
        x, y = a[i:i + chunk].float(), b[i:i + chunk].float()
        out[i:i + chunk] = (F.normalize(x, dim=-1) * F.normalize(y, dim=-1)).sum(-1)
    return out


def consecutive_run(accept: torch.Tensor, seq_lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Length of the all-accepted prefix of every row of a boolean ``[N, S]`` matrix."""
    run = accept.int().cumprod(dim=1).sum(dim=1).float()
    if seq_lens is not None:
        run = torch.minimum(run, seq_lens.to(run))
    return run


def _summary(values: torch.Tensor, prefix: str, suffix: str = "") -> Dict[str, float]:
    std = values.std().item() if values.numel() > 1 else 0.0
    return {f"{prefix}mean{suffix}": values.mean().item(), f"{prefix}std{suffix}": std,
            f"{prefix}max{suffix}": values.max().item(), f"{prefix}min{suffix}": values.min().item(),
            f"{prefix}median{suffix}": values.median().item()}


def sd_decode_time(t: TimingConfig, accepted: float, gamma: int, n_tokens: float = NUM_OUTPUT_TOKENS,
                   single_pass_verify: bool = False) -> float:
    """Analytic decode-phase time for ``n_tokens`` at ``accepted`` tokens per iteration.

    ``single_pass_verify=False`` prices a verification as gamma+1 target steps
    (measure_feature_acceptance.py:257-265); ``True`` prices it as one batched
    target forward (eval_two_phase.py:250-262), which is what the fused verify
    pass of the runtime actually does.
    """
    if accepted <= 0:
        return n_tokens * t.vl_per_token
    verify = t.vl_per_token if single_pass_verify else t.vl_per_token * (gamma + 1)
    return n_tokens / (accepted + 1) * (verify + t.adapter_latency)


# ------------------------------------------------------------ feature-level set
def metrics_from_cosine(cos_sim: torch.Tensor, mask: torch.Tensor, timing: Optional[TimingConfig] = None,
                        sd: Optional[SDConfig] = None, hidden_dim: int = 0) -> Dict:
    timing, sd = timing or TimingConfig(), sd or SDConfig()
    mb = mask.bool()
    valid = cos_sim[mb]
    n_valid = float(mb.sum())
    lens = mask.sum(dim=1)
    m: Dict = {"timestamp": datetime.now().isoformat(), "batch_size": cos_sim.shape[0],
               "max_seq_len": cos_sim.shape[1], "hidden_dim": hidden_dim,
               "total_valid_positions": int(n_valid)}
    m.update({k.replace("cs_", "cos_sim_"): v for k, v in _summary(valid, "cs_").items()})

    for th in sd.thresholds:
        key = int(round(th * 100))
        acc = cos_sim > th
        m[f"accept_{key}"] = float((acc & mb).sum()) / max(n_valid, 1.0)
        m.update(_summary(consecutive_run(acc, lens), "consecutive_", f"_{key}"))

    gamma_prefill = lens.float().mean().item()
    m["num_hidden_tokens_mean"] = gamma_prefill
    m["num_hidden_tokens_std"] = lens.float().std().item() if lens.numel() > 1 else 0.0
    m["num_hidden_tokens_min"] = lens.min().item()
    m["num_hidden_tokens_max"] = lens.max().item()

    consec_90 = m.get("consecutive_mean_90", 0.0)
    rate_prefill = consec_90 / gamma_prefill if gamma_prefill > 0 else 0.0
    g = sd.gamma_decode
    if cos_sim.shape[1] >= g:
        consec_g = consecutive_run((cos_sim[:, :g] > 0.90) & mb[:, :g]).mean().item()
        rate_decode = consec_g / g
    else:
        consec_g, rate_decode = consec_90, rate_prefill
    m.update({"sd_accept_rate_prefill_gamma": rate_prefill, "sd_gamma_prefill": gamma_prefill,
              "sd_accept_rate_decode_gamma5": rate_decode, "sd_consecutive_gamma5": consec_g,
              "sd_gamma_decode": g})

    t = timing
    free = t.free_draft_tokens
    effective = min(consec_90, free)
    m["speedup_prefill"] = 1 + effective * t.vl_per_token / t.vl_prefill if t.vl_prefill > 0 else 1.0
    m["prefill_overlap_window_ms"] = min(t.vl_prefill, t.egpt_prefill + t.adapter_latency)
    m["prefill_free_draft_tokens"] = free
    m["prefill_effective_tokens"] = effective
    overhead = t.adapter_latency / t.vl_per_token if t.vl_per_token > 0 else 0.1
    m["speedup_decode_gamma5"] = (consec_g + 1) / (1 + overhead)
    m["decode_overhead_ratio"] = overhead
    total = t.parallel_prefill + sd_decode_time(t, consec_g, g)
    m["speedup_e2e"] = t.baseline_ms() / total if total > 0 else 1.0
    m["e2e_baseline_time_ms"] = t.baseline_ms()
    m["e2e_sd_time_ms"] = total
    m["e2e_num_output_tokens"] = NUM_OUTPUT_TOKENS

    acc90, mean_sim = [], []
    for pos in range(min(cos_sim.shape[1], 20)):
        col = mb[:, pos]
        if not col.any():
            break
        s = cos_sim[col, pos]
        acc90.append((s > 0.90).float().mean().item())
        mean_sim.append(s.mean().item())
    m["position_accept_90"], m["position_mean_sim"] = acc90, mean_sim
    return m


def compute_all_metrics_parallel(aligned_hidden: torch.Tensor, vl_hidden: torch.Tensor, mask: torch.Tensor,
                                 timing_config: Optional[TimingConfig] = None,
                                 sd_config: Optional[SDConfig] = None):
    """``(metrics, cos_sim [N,S], valid_sims)`` — measure_feature_acceptance.py:60-294."""
    cos = masked_cosine(aligned_hidden, vl_hidden)
    mask = mask.to(cos.device)
    m = metrics_from_cosine(cos, mask, timing_config, sd_config, aligned_hidden.shape[-1])
    return m, cos, cos[mask.bool()]


def compute_per_position_stats(cos_sim: torch.Tensor, mask: torch.Tensor,
                               thresholds: Sequence[float] = (0.80, 0.85, 0.90, 0.95)) -> List[Dict]:
    out = []
    mb = mask.bool()
    for pos in range(cos_sim.shape[1]):
        col = mb[:, pos]
        n = int(col.sum())
        if n == 0:
            break
        s = cos_sim[col, pos]
        row = {"position": pos, "mean": s.mean().item(), "std": s.std().item() if n > 1 else 0.0,
               "num_samples": n}
        for th in thresholds:
            row[f"accept_{int(round(th * 100))}"] = (s > th).float().mean().item()
        out.append(row)
    return out


# -------------------------------------------------------------- token-level set
def load_lm_head(path: str) -> torch.Tensor:
    """``{'lm_head_weight': [V, H]}`` written by the LM-head extractor."""
    return torch.load(path, map_location="cpu", weights_only=True)["lm_head_weight"]


def project_to_tokens(hidden: torch.Tensor, lm_head_weight: torch.Tensor, mask: Optional[torch.Tensor] = None,
                      top_k: int = 10, batch_size: int = 64, device: Optional[torch.device] = None):
    """Greedy and top-k token ids of ``hidden @ W^T`` without materialising ``[N,S,V]``.

    On a GPU the top-1 path is the bf16 MFMA GEMM whose epilogue reduces fp32
    logits to per-tile (max, index) partials, so logits never reach HBM; top-k
    (k>1) falls back to a chunked fp32 GEMM + ``topk``. Returns CPU int64
    ``[N,S]`` and ``[N,S,k]``  (measure_feature_acceptance.py:693-733).
    """
    N, S, D = hidden.shape
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else hidden.device
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    W = lm_head_weight.to(device, torch.bfloat16 if on_gpu else torch.float32)
    W32 = None
    tok = torch.zeros(N, S, dtype=torch.long)
    topk = torch.zeros(N, S, top_k, dtype=torch.long)
    for i in range(0, N, batch_size):
        h = hidden[i:i + batch_size].to(device)
        rows = h.reshape(-1, D)
        if on_gpu and top_k == 1:
            from ..ops import hip
            pv, pi = hip.raw.linear_argmax_partials(rows.to(torch.bfloat16).contiguous(), W, None)
            ids = pi.gather(1, pv.argmax(1, keepdim=True))[:, 0].long()
            tok[i:i + batch_size] = ids.view(h.shape[:2]).cpu()
            topk[i:i + batch_size, :, 0] = tok[i:i + batch_size]
            continue
        if W32 is None:
            W32 = W.float()
        for j in range(0, rows.shape[0], 4096):
            logits = rows[j:j + 4096].float() @ W32.T
            tk = logits.topk(top_k, dim=-1).indices
            b0 = i * S + j
            topk.view(-1, top_k)[b0:b0 + tk.shape[0]] = tk.cpu()
        tok[i:i + batch_size] = topk[i:i + batch_size, :, 0]
    return tok, topk


def token_metrics_from_ids(aligned_tokens: torch.Tensor, aligned_topk: torch.Tensor,
                           vl_tokens: torch.Tensor, mask: torch.Tensor) -> Dict:
    mb = mask.bool().cpu()
    n_valid = max(float(mb.sum()), 1.0)
    top1 = (aligned_tokens == vl_tokens) & mb
    vl_e = vl_tokens.unsqueeze(-1)
    m = {"token_match_rate": float(top1.sum()) / n_valid,
         "token_match_rate_top5": float(((aligned_topk[:, :, :5] == vl_e).any(-1) & mb).sum()) / n_valid,
         "token_match_rate_top10": float(((aligned_topk == vl_e).any(-1) & mb).sum()) / n_valid}
    m.update(_summary(consecutive_run(top1, mask.sum(dim=1).cpu()), "token_consecutive_"))
    per_pos = []
    for pos in range(min(top1.shape[1], 20)):
        col = mb[:, pos]
        if not col.any():
            break
        per_pos.append(top1[col, pos].float().mean().item())
    m["position_token_accept"] = per_pos
    return m


def compute_token_level_metrics(aligned_hidden, vl_hidden, mask, lm_head_weight, top_k: int = 10,
                                batch_size: int = 64):
    """``(metrics, aligned_tokens, vl_tokens)`` — measure_feature_acceptance.py:736-802."""
    a_tok, a_topk = project_to_tokens(aligned_hidden, lm_head_weight, mask, top_k, batch_size)
    v_tok, _ = project_to_tokens(vl_hidden, lm_head_weight, mask, 1, batch_size)
    return token_metrics_from_ids(a_tok, a_topk, v_tok, mask), a_tok, v_tok


def compute_two_phase_sd_metrics(aligned_tokens, vl_tokens, mask, timing_config: Optional[TimingConfig] = None,
                                 gamma_decode: int = 5) -> Dict:
    """Token-level prefill-hiding + decode estimates (measure_feature_acceptance.py:805-879)."""
    t = timing_config or TimingConfig()
    mb = mask.bool().cpu()
    match = (aligned_tokens == vl_tokens) & mb
    S = match.shape[1]
    gp = int(t.free_draft_tokens)
    m: Dict = {"token_sd_gamma_prefill": gp}
    if gp > 0:
        first = min(gp, S)
        mean = consecutive_run(match[:, :first]).mean().item()
        m["token_sd_prefill_consecutive_mean"] = mean
        m["token_sd_prefill_accept_rate"] = mean / first
    else:
        m["token_sd_prefill_consecutive_mean"] = 0.0
        m["token_sd_prefill_accept_rate"] = 0.0
    accepted = consecutive_run(match[:, :gamma_decode]).mean().item() if S >= gamma_decode else 0.0
    m["token_sd_decode_consecutive_mean"] = accepted
    m["token_sd_decode_accept_rate"] = accepted / gamma_decode if gamma_decode > 0 else 0.0
    m["token_sd_gamma_decode"] = gamma_decode
    total = t.parallel_prefill + sd_decode_time(t, accepted, gamma_decode)
    m["token_sd_e2e_speedup"] = t.baseline_ms() / total if total > 0 else 1.0
    m["token_sd_e2e_baseline_ms"] = t.baseline_ms()
    m["token_sd_e2e_sd_ms"] = total
    return m


def compute_cosim_vs_token_correlation(cos_sim: torch.Tensor, token_match: torch.Tensor,
                                       mask: torch.Tensor) -> Dict:
    """How well a cosine threshold predicts a token match (measure_feature_acceptance.py:882-915)."""
    cos_sim, token_match = cos_sim.cpu(), token_match.cpu().bool()
    mb = mask.bool().cpu()
    out: Dict = {}
    for th in CORRELATION_THRESHOLDS:
        above = (cos_sim > th) & mb
        n = int(above.sum())
        key = int(round(th * 100))
        out[f"cosim_{key}_token_match"] = float((token_match & above).sum()) / n if n else 0.0
        out[f"cosim_{key}_count"] = n
    cs, tm = cos_sim[mb].float(), token_match[mb].float()
    r = 0.0
    if cs.numel() > 1 and cs.std() > 0 and tm.std() > 0:
        r = torch.corrcoef(torch.stack([cs, tm]))[0, 1].item()
    out["cosim_token_pearson_r"] = r
    return out


# ----------------------------------------------------------------- stage model
def stage_breakdown(metrics: Dict, timing: TimingConfig) -> Dict[str, float]:
    """Numbers behind the stage-timeline figure (measure_feature_acceptance.py:541-552)."""
    n = metrics.get("e2e_num_output_tokens", NUM_OUTPUT_TOKENS)
    acc, g = metrics["sd_consecutive_gamma5"], metrics["sd_gamma_decode"]
    decode = sd_decode_time(timing, acc, g, n)
    total = timing.parallel_prefill + decode
    return {"baseline_vl_prefill_ms": timing.vl_prefill, "baseline_vl_decode_ms": n * timing.vl_per_token,
            "baseline_total_ms": timing.baseline_ms(n), "sd_parallel_prefill_ms": timing.parallel_prefill,
            "sd_decode_ms": decode, "sd_total_ms": total,
            "sd_iterations": n / (acc + 1) if acc > 0 else n,
            "time_ratio_prefill": timing.parallel_prefill / total, "time_ratio_decode": decode / total}


# ------------------------------------------------------------------- data + CLI
def load_eval_data(path: str, device: str = "cpu"):
    """``(egpt_hidden, vl_hidden, mask)`` from a chunked directory or a single ``.pt``."""
    from ..align.store import load_hidden_states
    d = load_hidden_states(str(path))
    e, v = d["egpt_hidden"], d["vl_hidden"]
    lens = torch.as_tensor(d["seq_lens"])
    return e.to(device), v.to(device), seq_mask(lens, e.shape[1]).to(device)


@torch.no_grad()
def apply_adapter(adapter, egpt_hidden, vl_hidden, batch_size: int = 64, device="cpu",
                  vlm_only: bool = False) -> torch.Tensor:
    """Run any adapter level over the set in chunks; fused adapters see both streams."""
    fused = type(adapter).__name__ == "FusedEAGLEAdapter"
    dtype = next(adapter.parameters()).dtype
    outs = []
    for i in range(0, egpt_hidden.shape[0], batch_size):
        e = egpt_hidden[i:i + batch_size].to(device, dtype)
        v = vl_hidden[i:i + batch_size].to(device, dtype)
        out = adapter(e, v) if fused else adapter(v if vlm_only else e)
        outs.append(out.float().cpu())
    return torch.cat(outs)


def evaluate(aligned: torch.Tensor, vl_hidden: torch.Tensor, mask: torch.Tensor,
             timing: Optional[TimingConfig] = None, sd: Optional[SDConfig] = None,
             lm_head_weight: Optional[torch.Tensor] = None, batch_size: int = 64) -> Tuple[Dict, Dict]:
    """All metric groups in one dict plus the tensors plots need."""
    timing, sd = timing or TimingConfig(), sd or SDConfig()
    metrics, cos, valid = compute_all_metrics_parallel(aligned, vl_hidden, mask, timing, sd)
    extras = {"cos_sim": cos, "valid_sims": valid,
              "position_stats": compute_per_position_stats(cos, mask, sd.thresholds)}
    metrics["stage_breakdown"] = stage_breakdown(metrics, timing)
    if lm_head_weight is not None:
        tm, a_tok, v_tok = compute_token_level_metrics(aligned, vl_hidden, mask, lm_head_weight,
                                                       batch_size=batch_size)
        metrics.update(tm)
        metrics.update(compute_two_phase_sd_metrics(a_tok, v_tok, mask, timing, sd.gamma_decode))
        metrics.update(compute_cosim_vs_token_correlation(cos, (a_tok == v_tok), mask))
        extras.update(aligned_tokens=a_tok, vl_tokens=v_tok)
    metrics["timing_config"] = asdict(timing)
    return metrics, extras


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="Feature/token-level SD acceptance metrics")
    p.add_argument("--checkpoint", required=True, help="adapter checkpoint (any level)")
    p.add_argument("--test_data", required=True, help="chunked hidden-state directory or .pt")
    p.add_argument("--output_dir", default=None)
    p.add_argument("--lm_head", default=None, help="enable token-level metrics with these LM-head weights")
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--vlm_only", action="store_true")
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--no_plots", action="store_true")
    for f, d in asdict(TimingConfig()).items():
        p.add_argument(f"--{f}_ms", type=float, default=d)
    p.add_argument("--gamma_decode", type=int, default=5)
    return p


def main(argv=None) -> Dict:
    from ..align.adapters import load_any_adapter
    from . import reports
    args = build_parser().parse_args(argv)
    timing = TimingConfig(**{f: getattr(args, f + "_ms") for f in asdict(TimingConfig())})
    sd = SDConfig(gamma_decode=args.gamma_decode)
    adapter, _ = load_any_adapter(args.checkpoint, args.device)
    adapter = adapter.to(args.device).eval()
    e, v, mask = load_eval_data(args.test_data)
    aligned = apply_adapter(adapter, e, v, args.batch_size, args.device, args.vlm_only)
    W = load_lm_head(args.lm_head) if args.lm_head else None
    metrics, extras = evaluate(aligned, v, mask, timing, sd, W, args.batch_size)
    metrics["adapter_class"] = type(adapter).__name__
    out_dir = args.output_dir or os.path.join(os.path.dirname(os.path.abspath(args.checkpoint)), "acceptance")
    os.makedirs(out_dir, exist_ok=True)
    reports.print_metrics_report(metrics)
    if W is not None:
        reports.print_token_metrics_report(metrics)
    reports.write_metrics(metrics, extras["position_stats"], out_dir)
    if not args.no_plots:
        reports.plot_metrics(metrics, extras["valid_sims"], extras["position_stats"], out_dir)
        reports.plot_stage_timeline(metrics, timing, out_dir)
        reports.plot_token_metrics(metrics, out_dir)
    return metrics


if __name__ == "__main__":
    main()
