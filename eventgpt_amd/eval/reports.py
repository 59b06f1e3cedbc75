"""Console / JSON / Markdown / PNG writers for evaluation and benchmark results.

Output file names follow the reference (``acceptance_metrics.json``,
``position_stats.json``, ``metrics_summary.png``, ``stage_timeline.png``,
``token_metrics.png``, ``metrics_comparison.md`` —
pipeline/evaluation/measure_feature_acceptance.py:327-630, 1060-1110, 1276-1315).
Plotting is optional: without matplotlib the plot functions return ``None``.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional, Sequence

import torch

_THRESH = (80, 85, 90, 95)


def _bar(rate: float, width: int = 20) -> str:
    return "#" * int(max(0.0, min(1.0, rate)) * width)


def to_jsonable(obj):
    if isinstance(obj, torch.Tensor):
        return obj.item() if obj.numel() == 1 else obj.tolist()
    if isinstance(obj, dict):
        return {str(k): to_jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_jsonable(v) for v in obj]
    if hasattr(obj, "item") and not isinstance(obj, (str, bytes)):
        try:
            return obj.item()
        except Exception:
            return str(obj)
    if isinstance(obj, float) and obj != obj:
        return None
    return obj


def write_json(obj, path: str) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(to_jsonable(obj), f, indent=2)
    os.replace(tmp, path)
    return path


def markdown_table(headers: Sequence[str], rows: Iterable[Sequence]) -> str:
    def fmt(v):
        return f"{v:.4g}" if isinstance(v, float) else str(v)
    out = ["| " + " | ".join(headers) + " |", "|" + "|".join("---" for _ in headers) + "|"]
    out += ["| " + " | ".join(fmt(v) for v in r) + " |" for r in rows]
    return "\n".join(out)


# ----------------------------------------------------------------------- console
def print_metrics_report(m: Dict, out=print) -> None:
    line = "=" * 78
    out(line)
    out("FEATURE-LEVEL SPECULATIVE DECODING METRICS")
    out(f"generated {m.get('timestamp', '')}")
    out(line)
    out(f"samples {m['batch_size']}  max_seq {m['max_seq_len']}  hidden {m['hidden_dim']}  "
        f"valid positions {m['total_valid_positions']}")
    out(f"cosine  mean {m['cos_sim_mean']:.4f}  std {m['cos_sim_std']:.4f}  median {m['cos_sim_median']:.4f}  "
        f"min {m['cos_sim_min']:.4f}  max {m['cos_sim_max']:.4f}")
    out("acceptance / consecutive accepts by threshold")
    for th in _THRESH:
        if f"accept_{th}" not in m:
            continue
        out(f"  @0.{th}: {m[f'accept_{th}']:7.2%} {_bar(m[f'accept_{th}']):20s} "
            f"run {m[f'consecutive_mean_{th}']:5.2f} +- {m[f'consecutive_std_{th}']:4.2f} "
            f"(max {m[f'consecutive_max_{th}']:.0f})")
    out(f"hidden tokens / sample: {m['num_hidden_tokens_mean']:.2f} +- {m['num_hidden_tokens_std']:.2f} "
        f"[{m['num_hidden_tokens_min']:.0f}, {m['num_hidden_tokens_max']:.0f}]")
    out(f"prefill SD (gamma={m['sd_gamma_prefill']:.1f}): accept rate {m['sd_accept_rate_prefill_gamma']:.2%}")
    out(f"decode  SD (gamma={m['sd_gamma_decode']}): run {m['sd_consecutive_gamma5']:.2f}, "
        f"accept rate {m['sd_accept_rate_decode_gamma5']:.2%}")
    out(f"speedup  prefill {m['speedup_prefill']:.2f}x (free draft tokens {m['prefill_free_draft_tokens']:.1f}, "
        f"effective {m['prefill_effective_tokens']:.1f})  decode {m['speedup_decode_gamma5']:.2f}x  "
        f"e2e {m['speedup_e2e']:.2f}x ({m['e2e_baseline_time_ms']:.0f} -> {m['e2e_sd_time_ms']:.0f} ms)")
    out("per-position acceptance @0.90")
    for i, r in enumerate(m.get("position_accept_90", [])[:10]):
        out(f"  pos {i:2d}: {r:7.2%} {_bar(r)}")
    out(line)


def print_token_metrics_report(m: Dict, out=print) -> None:
    line = "=" * 78
    out(line)
    out("TOKEN-LEVEL METRICS (LM-head projection)")
    out(line)
    out(f"top-1 {m['token_match_rate']:.2%}  top-5 {m['token_match_rate_top5']:.2%}  "
        f"top-10 {m['token_match_rate_top10']:.2%}")
    out(f"consecutive matches {m['token_consecutive_mean']:.2f} +- {m['token_consecutive_std']:.2f} "
        f"(median {m['token_consecutive_median']:.0f}, max {m['token_consecutive_max']:.0f})")
    if "token_sd_gamma_prefill" in m:
        out(f"prefill hiding gamma={m['token_sd_gamma_prefill']}: run {m['token_sd_prefill_consecutive_mean']:.2f} "
            f"rate {m['token_sd_prefill_accept_rate']:.2%}")
        out(f"decode gamma={m['token_sd_gamma_decode']}: run {m['token_sd_decode_consecutive_mean']:.2f} "
            f"rate {m['token_sd_decode_accept_rate']:.2%}  e2e {m['token_sd_e2e_speedup']:.2f}x")
    if "cosim_token_pearson_r" in m:
        out(f"cosine vs token match: pearson r = {m['cosim_token_pearson_r']:.3f}")
        for th in (80, 85, 90, 92, 95, 98):
            if f"cosim_{th}_count" in m:
                out(f"  cos>0.{th}: token match {m[f'cosim_{th}_token_match']:.2%} over {m[f'cosim_{th}_count']}")
    for i, r in enumerate(m.get("position_token_accept", [])[:10]):
        out(f"  pos {i:2d}: {r:7.2%} {_bar(r)}")
    out(line)


# ------------------------------------------------------------------------- files
def write_metrics(metrics: Dict, position_stats: List[Dict], out_dir: str) -> Dict[str, str]:
    paths = {"metrics": write_json(metrics, os.path.join(out_dir, "acceptance_metrics.json")),
             "positions": write_json(position_stats, os.path.join(out_dir, "position_stats.json"))}
    if "token_match_rate" in metrics:
        paths["comparison"] = write_metrics_comparison_md(metrics, out_dir)
    return paths


def write_metrics_comparison_md(m: Dict, out_dir: str) -> str:
    """Feature-level vs token-level side by side."""
    rows = [("overall accept", f"{m.get('accept_90', 0):.2%} (cos>0.90)", f"{m['token_match_rate']:.2%} (top-1)"),
            ("consecutive mean", f"{m.get('consecutive_mean_90', 0):.2f}", f"{m['token_consecutive_mean']:.2f}"),
            (f"decode run (gamma={m.get('sd_gamma_decode', 5)})", f"{m.get('sd_consecutive_gamma5', 0):.2f}",
             f"{m.get('token_sd_decode_consecutive_mean', 0):.2f}"),
            ("e2e speedup", f"{m.get('speedup_e2e', 0):.2f}x", f"{m.get('token_sd_e2e_speedup', 0):.2f}x")]
    text = ["# Feature-level vs token-level acceptance", "",
            markdown_table(["metric", "feature level", "token level"], rows), "",
            f"Top-5 / top-10 token match: {m['token_match_rate_top5']:.2%} / {m['token_match_rate_top10']:.2%}", ""]
    if "cosim_token_pearson_r" in m:
        text += ["## Is cosine similarity a proxy for token acceptance?", "",
                 markdown_table(["cos threshold", "positions above", "token match among them"],
                                [(f"0.{th}", m[f"cosim_{th}_count"], f"{m[f'cosim_{th}_token_match']:.2%}")
                                 for th in (80, 85, 90, 92, 95, 98) if f"cosim_{th}_count" in m]),
                 "", f"Pearson r = {m['cosim_token_pearson_r']:.3f}", ""]
    path = os.path.join(out_dir, "metrics_comparison.md")
    with open(path, "w") as f:
        f.write("\n".join(text))
    return path


# ------------------------------------------------------------------------- plots
def _plt():
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except Exception:
        return None


def plot_metrics(m: Dict, cos_sim_flat: torch.Tensor, position_stats: List[Dict], out_dir: str) -> Optional[str]:
    plt = _plt()
    if plt is None:
        return None
    fig, ax = plt.subplots(2, 2, figsize=(13, 9))
    colors = dict(zip(_THRESH, ("tab:green", "tab:orange", "tab:red", "tab:purple")))
    ax[0, 0].hist(cos_sim_flat.float().cpu().numpy(), bins=50, color="steelblue", alpha=0.8)
    for th, c in colors.items():
        ax[0, 0].axvline(th / 100, ls="--", color=c, label=f"tau=0.{th}")
    ax[0, 0].set(xlabel="cosine similarity", ylabel="positions", title="Hidden-state cosine distribution")
    ax[0, 0].legend()
    head = position_stats[:20]
    for th, c in colors.items():
        ax[0, 1].plot([p["position"] for p in head], [p.get(f"accept_{th}", 0) for p in head], "o-", color=c,
                      label=f"@0.{th}")
    ax[0, 1].set(xlabel="position", ylabel="acceptance", ylim=(0, 1.05), title="Acceptance by position")
    ax[0, 1].legend()
    means = [m.get(f"consecutive_mean_{th}", 0) for th in _THRESH]
    ax[1, 0].bar(range(4), means, yerr=[m.get(f"consecutive_std_{th}", 0) for th in _THRESH], capsize=4,
                 color=list(colors.values()), alpha=0.8)
    ax[1, 0].set_xticks(range(4))
    ax[1, 0].set_xticklabels([f"0.{th}" for th in _THRESH])
    ax[1, 0].set(ylabel="tokens", title="Consecutive accepts by threshold")
    names = ["prefill", f"decode g={m.get('sd_gamma_decode', 5)}", "e2e"]
    vals = [m["speedup_prefill"], m["speedup_decode_gamma5"], m["speedup_e2e"]]
    bars = ax[1, 1].bar(names, vals, color=["steelblue", "coral", "seagreen"], alpha=0.8)
    ax[1, 1].axhline(1.0, color="gray", ls="--")
    for b, v in zip(bars, vals):
        ax[1, 1].text(b.get_x() + b.get_width() / 2, v, f"{v:.2f}x", ha="center", va="bottom")
    ax[1, 1].set(ylabel="speedup", title="Estimated speedup")
    for a in ax.flat:
        a.grid(alpha=0.3)
    fig.tight_layout()
    path = os.path.join(out_dir, "metrics_summary.png")
    fig.savefig(path, dpi=120)
    plt.close(fig)
    return path


def plot_stage_timeline(m: Dict, timing, out_dir: str) -> Optional[str]:
    """Baseline (prefill -> decode) against SD (parallel prefill -> SD decode) as stacked bars."""
    plt = _plt()
    if plt is None:
        return None
    sb = m.get("stage_breakdown")
    if sb is None:
        from .acceptance import stage_breakdown
        sb = stage_breakdown(m, timing)
    fig, ax = plt.subplots(figsize=(12, 3.8))
    ax.barh(1.0, sb["baseline_vl_prefill_ms"], color="tab:blue", edgecolor="k", label="verifier prefill")
    ax.barh(1.0, sb["baseline_vl_decode_ms"], left=sb["baseline_vl_prefill_ms"], color="tab:green", edgecolor="k",
            label="verifier decode")
    ax.barh(0.15, timing.vl_prefill, height=0.3, color="tab:blue", edgecolor="k")
    ax.barh(-0.15, timing.egpt_prefill, height=0.3, color="tab:red", edgecolor="k", label="drafter prefill")
    ax.barh(-0.15, timing.adapter_latency, left=timing.egpt_prefill, height=0.3, color="tab:purple",
            edgecolor="k", label="adapter")
    ax.barh(0.0, sb["sd_decode_ms"], left=sb["sd_parallel_prefill_ms"], height=0.6, color="tab:orange",
            edgecolor="k", label=f"SD decode ({sb['sd_iterations']:.1f} iters)")
    ax.set_yticks([0.0, 1.0])
    ax.set_yticklabels(["with SD", "baseline"])
    ax.set_xlabel("ms")
    ax.set_title(f"baseline {sb['baseline_total_ms']:.0f} ms vs SD {sb['sd_total_ms']:.0f} ms "
                 f"({sb['baseline_total_ms'] / sb['sd_total_ms']:.2f}x)")
    ax.legend(loc="upper right", fontsize=8)
    ax.grid(alpha=0.3, axis="x")
    fig.tight_layout()
    path = os.path.join(out_dir, "stage_timeline.png")
    fig.savefig(path, dpi=120)
    plt.close(fig)
    return path


def plot_token_metrics(m: Dict, out_dir: str) -> Optional[str]:
    plt = _plt()
    if plt is None or "token_match_rate" not in m:
        return None
    fig, ax = plt.subplots(1, 3, figsize=(15, 4.2))
    ax[0].bar(["top-1", "top-5", "top-10"],
              [m["token_match_rate"], m["token_match_rate_top5"], m["token_match_rate_top10"]],
              color=["tab:red", "tab:orange", "tab:green"], alpha=0.8)
    ax[0].set(ylim=(0, 1.05), title="token match rate")
    pos = m.get("position_token_accept", [])
    ax[1].plot(range(len(pos)), pos, "o-", label="token top-1")
    if m.get("position_accept_90"):
        ax[1].plot(range(len(m["position_accept_90"])), m["position_accept_90"], "s--", label="cos>0.90")
    ax[1].set(ylim=(0, 1.05), xlabel="position", title="per-position acceptance")
    ax[1].legend()
    ths = [th for th in (80, 85, 90, 92, 95, 98) if f"cosim_{th}_token_match" in m]
    ax[2].plot([th / 100 for th in ths], [m[f"cosim_{th}_token_match"] for th in ths], "o-")
    ax[2].set(ylim=(0, 1.05), xlabel="cosine threshold", title="P(token match | cos > threshold)")
    for a in ax:
        a.grid(alpha=0.3)
    fig.tight_layout()
    path = os.path.join(out_dir, "token_metrics.png")
    fig.savefig(path, dpi=120)
    plt.close(fig)
    return path
