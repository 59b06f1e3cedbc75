"""End-to-end wall-clock speculative-decoding benchmark: drafter (EventGPT) +
hidden-state adapters against a Video-LLaVA-shaped verifier, every method and
baseline measured on the same samples.

Counterpart of the reference's serving-path benchmark
(pipeline/benchmark_e2e/benchmark_e2e_wallclock.py:59-2204): 3-stage timing of
both models, adapter discovery, the config families

    L?          adapter-drafted SD with a fixed gamma
    B1 / L5F    gamma=1 feedback drafting from verifier (+ drafter) hidden states
    L?+L5F      prefill hiding with L? free tokens, then L5F decode
    L?+VL       prefill hiding, then plain verifier decode

and a JSON + Markdown + PNG report (``e2e_wallclock_<ts>.*``).  What differs is the
machinery underneath: both models run on graph-captured engines; verification is
one forward + a fused accept kernel; prefill hiding is *live* (the drafter really
decodes on its own stream while the verifier prefills, instead of a simulation
from separately measured times); every SD output is checked token-for-token
against the verifier's own greedy decode.
"""
from __future__ import annotations

import argparse
import json
import os
import time
from datetime import datetime
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import ops
from ..align.adapters import load_any_adapter
from ..constants import EVENT_TOKEN_INDEX
from ..data.conversation import prepare_event_prompt
from ..data.events import load_image
from ..data.tokenizer import tokenizer_event_token
from ..eval.reports import markdown_table, write_json
from ..eval.timers import StageTimer
from ..models.event_chat import get_spatio_temporal_features
from ..models.video_llava import VIDEO_PROMPT, load_video_frames
from ..pipeline.dataset import InstructionDataset, load_questions
from ..runtime.scheduler import DualModelScheduler
from ..runtime.spec_decode import (AdapterDraftHead, SDStats, SpecDecodeEngine, collect_decode_hidden,
                                   matches_greedy)

LEVELS = ("L1", "L2", "L3", "L4", "L5", "B1", "L5F")
PREFILL_LEVELS = ("L1", "L2", "L3", "L4", "L5")


# ------------------------------------------------------------ sample preparation
class SampleInputs:
    """Host-side preprocessing of one (scene, question): pixel tensors and token ids."""

    def __init__(self, egpt, egpt_tok, vl, vl_proc, event_image_key: str = "event_image"):
        self.egpt, self.egpt_tok, self.vl, self.vl_proc = egpt, egpt_tok, vl, vl_proc
        self.key = event_image_key

    def load(self, ds: InstructionDataset, sample: Dict, query: str) -> Optional[Dict]:
        paths = ds.event_image_paths(sample, self.key)
        src = ds.video_source(sample)
        if not paths or not os.path.exists(paths[0]) or src is None:
            return None
        frames = load_video_frames(src, self.vl.config.num_frames)
        if not frames:
            return None
        img = np.asarray(load_image(paths[0]))
        from ..ops.event_ops import preprocess_frames
        ev_px = preprocess_frames([img], self.egpt.device, self.egpt.get_visual_tower().event_processor.size,
                                  self.egpt.dtype)[0]
        ev_ids = tokenizer_event_token(prepare_event_prompt(query, "eventgpt_v1"), self.egpt_tok,
                                       EVENT_TOKEN_INDEX, return_tensors="pt")[None]
        vin = self.vl_proc(VIDEO_PROMPT.format(query=query), frames, device=self.vl.device, dtype=self.vl.dtype)
        return {"ev_px": ev_px, "ev_ids": ev_ids, "ev_size": list(img.shape[:2]),
                "vl_ids": vin["input_ids"], "vl_px": vin["pixel_values_videos"]}

    @torch.no_grad()
    def egpt_embeds(self, inp: Dict) -> torch.Tensor:
        m = self.egpt
        feats = m.encode_event_frames([inp["ev_px"]])[None]
        return m.prepare_inputs_labels_for_multimodal(
            inp["ev_ids"].to(m.device), None, None, None, None, event_features=feats,
            event_image_sizes=[inp["ev_size"]], already_adapted=True)[4][0]

    @torch.no_grad()
    def vl_embeds(self, inp: Dict) -> torch.Tensor:
        m = self.vl
        return m.prepare_embeds(inp["vl_ids"].to(m.device), inp["vl_px"].to(m.device, m.dtype))[0]


# --------------------------------------------------------------- 3-stage timing
@torch.no_grad()
def time_3stage(kind: str, prep: SampleInputs, inp: Dict, max_new_tokens: int, eos: int) -> Dict[str, float]:
    """vision / prefill / decode of one model on its engine (ms)."""
    model = prep.egpt if kind == "egpt" else prep.vl
    eng = model.engine()
    eng.eos = eos
    st = StageTimer(model.device)
    with st.stage("vision_ms"):
        emb = prep.egpt_embeds(inp) if kind == "egpt" else prep.vl_embeds(inp)
    with st.stage("prefill_ms"):
        eng.reset()
        eng.prefill(emb, select="greedy")
    with st.stage("decode_ms"):
        toks = eng.generate_continue(max_new_tokens, eos)
    r = st.result()
    r["total_ms"] = r["vision_ms"] + r["prefill_ms"] + r["decode_ms"]
    r["num_tokens"] = int(toks.shape[0])
    r["prefill_len"] = int(emb.shape[0])
    return r


# ----------------------------------------------------------- adapter discovery
def find_adapter_checkpoints(base_dir: Optional[str]) -> Dict[str, str]:
    """``{level: best_model.pt}``.  Accepts ``<base>/<L?>/<L?>_<ts>/best_model.pt``
    (latest timestamp wins), ``<base>/<L?>_<ts>/best_model.pt``, or a run directory
    itself (benchmark_e2e_wallclock.py:1039-1094)."""
    found: Dict[str, str] = {}
    if not base_dir or not os.path.isdir(base_dir):
        return found
    for level in LEVELS:
        ldir = os.path.join(base_dir, level)
        if not os.path.isdir(ldir):
            continue
        runs = sorted((d for d in os.listdir(ldir)
                       if os.path.exists(os.path.join(ldir, d, "best_model.pt"))), reverse=True)
        if runs:
            found[level] = os.path.join(ldir, runs[0], "best_model.pt")
        elif os.path.exists(os.path.join(ldir, "best_model.pt")):
            found[level] = os.path.join(ldir, "best_model.pt")
    if found:
        return found
    direct = os.path.join(base_dir, "best_model.pt")
    if os.path.exists(direct):
        level = os.path.basename(os.path.normpath(base_dir)).split("_")[0]
        return {level: direct} if level in LEVELS else {}
    for d in sorted(os.listdir(base_dir), reverse=True):
        level = d.split("_")[0]
        if level in LEVELS and os.path.exists(os.path.join(base_dir, d, "best_model.pt")):
            found.setdefault(level, os.path.join(base_dir, d, "best_model.pt"))
    return found


def build_sd_configs(levels: List[str], only: Optional[List[str]] = None) -> List[Dict]:
    cfgs: List[Dict] = []
    for lv in levels:
        if lv in PREFILL_LEVELS:
            cfgs.append({"name": lv, "kind": "sd", "prefill": None, "decode": lv, "mode": "adapter"})
    for lv, mode in (("B1", "vlm_only"), ("L5F", "fused")):
        if lv in levels:
            cfgs.append({"name": lv, "kind": "sd", "prefill": None, "decode": lv, "mode": mode})
    if "L5F" in levels:
        for lv in levels:
            if lv in PREFILL_LEVELS:
                cfgs.append({"name": f"{lv}+L5F", "kind": "two_phase", "prefill": lv, "decode": "L5F",
                             "mode": "fused"})
    for lv in levels:
        if lv in PREFILL_LEVELS:
            cfgs.append({"name": f"{lv}+VL", "kind": "hiding", "prefill": lv, "decode": None, "mode": None})
    if only:
        cfgs = [c for c in cfgs if c["name"] in only]
    return cfgs


# ------------------------------------------------------------------ the runner
class E2EBenchmark:
    def __init__(self, egpt, egpt_tok, vl, vl_proc, adapters: Dict[str, torch.nn.Module], gamma: int = 5,
                 max_new_tokens: int = 50, event_image_key: str = "event_image", max_free: int = 30):
        self.prep = SampleInputs(egpt, egpt_tok, vl, vl_proc, event_image_key)
        self.egpt, self.vl = egpt, vl
        self.d, self.v = egpt.engine(), vl.engine()
        self.gamma, self.max_new, self.max_free = gamma, max_new_tokens, max_free
        self.eos = vl.config.text_config.eos_token_id
        w = vl.lm_head.weight.data
        self.heads = {k: AdapterDraftHead(a.to(vl.device).to(torch.bfloat16), w) for k, a in adapters.items()}
        self.sched = DualModelScheduler(self.d, self.v)
        self._sd: Dict[str, SpecDecodeEngine] = {}     # per draft head: keeps its captured rounds across samples

    # -- losslessness -------------------------------------------------------------
    NEAR_TIE = 0.08      # logit gap below which bf16 kernels of different shapes may order two tokens either way

    def _lossless(self, toks: torch.Tensor, ref: torch.Tensor, v_emb: torch.Tensor) -> Dict:
        """Token-identical to the verifier's greedy answer -- or first differing at a position where
        the verifier's own (teacher-forced) logits hold the two candidates closer than bf16 rounding:
        the verify pass scores several rows with other kernels than the one-row decode step, so such
        a tie may fall either way.  Scoring resets the verifier engine."""
        return matches_greedy(self.v, v_emb, toks, ref, self.NEAR_TIE)

    # -- one verifier prefill whose state the configs restart from ---------------
    def _verifier_prefill(self, v_emb: torch.Tensor):
        v = self.v
        v.eos = self.eos
        v.reset()
        h = v.prefill(v_emb, select="greedy", return_hidden=True)
        return ops.rmsnorm(h[-1:], v.final_norm, v.eps)

    def _restart(self, state) -> None:
        """Back to the post-prefill state: KV rollback is a position reset, the cache
        rows past it are simply overwritten (no 32x2 tensor clone as in the reference)."""
        v = self.v
        v.pos.copy_(state["pos"]); v.step.copy_(state["step"]); v.done.copy_(state["done"])
        v.cur_token.copy_(state["cur"]); v.out_tokens[:, :1].copy_(state["tok0"])
        v.x[0].copy_(v.embed[state["cur"][0].long()])

    def _snapshot(self) -> Dict:
        v = self.v
        return {"pos": v.pos.clone(), "step": v.step.clone(), "done": v.done.clone(),
                "cur": v.cur_token.clone(), "tok0": v.out_tokens[:, :1].clone()}

    @torch.no_grad()
    def run_sample(self, inp: Dict, configs: List[Dict]) -> Dict:
        prep = self.prep
        d_emb, v_emb = prep.egpt_embeds(inp), prep.vl_embeds(inp)
        out: Dict = {"prefill_len_egpt": int(d_emb.shape[0]), "prefill_len_vl": int(v_emb.shape[0])}
        base = self.sched.verifier_baseline(v_emb, self.max_new, self.eos)
        ref = base["tokens"]
        out["baseline"] = {"prefill_ms": base["prefill_ms"], "decode_ms": base["decode_ms"],
                           "total_ms": base["total_ms"], "tokens": int(ref.shape[0])}
        pp = self.sched.parallel_prefill(d_emb, v_emb)
        out["parallel_prefill"] = pp
        # the drafter's own answer + hidden states; its decode time is charged to the
        # configs that consume those states
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dh = collect_decode_hidden(self.d, d_emb, self.max_new, self.eos)
        torch.cuda.synchronize()
        egpt_decode_ms = max((time.perf_counter() - t0) * 1e3 - pp["egpt_ms"], 0.0)
        out["egpt_decode_ms"] = egpt_decode_ms
        out["configs"] = {}
        state = vh0 = None
        for cfg in configs:
            if cfg["kind"] == "sd":
                if state is None:
                    vh0 = self._verifier_prefill(v_emb)
                    state = self._snapshot()
                else:
                    self._restart(state)
                sd = self._sd.get(cfg["decode"])
                if sd is None:
                    sd = self._sd[cfg["decode"]] = SpecDecodeEngine(self.v, self.heads[cfg["decode"]], self.gamma)
                toks, st = sd.run(dh["hidden"], self.max_new, self.eos, cfg["mode"], vh0)
                extra = 0.0 if cfg["mode"] == "vlm_only" else egpt_decode_ms
                r = self._result(toks, st, pp["wall_ms"], extra, pp)
            else:
                r = self._hiding(cfg, d_emb, v_emb, dh["hidden"], egpt_decode_ms)
                state = None            # the verifier was prefilled again by the hiding run
            toks = r.pop("tokens")
            check = self._lossless(toks, ref, v_emb)
            if not check["exact"]:
                state = None            # scoring the mismatch reset the verifier
            r.update(check)
            r["speedup"] = base["total_ms"] / r["sd_total_ms"] if r["sd_total_ms"] > 0 else 0.0
            out["configs"][cfg["name"]] = r
        return out

    def _hiding(self, cfg, d_emb, v_emb, d_hidden, egpt_decode_ms) -> Dict:
        """Live prefill hiding: the drafter decodes on its stream while the verifier
        prefills; the free tokens are verified in one batch; then either the verifier
        continues on its own (``+VL``) or a gamma=1 fused drafter takes over (``+L5F``)."""
        sch = self.sched
        sch.head = self.heads[cfg["prefill"]]
        two_phase = cfg["kind"] == "two_phase"
        pre = sch.hidden_prefill(d_emb, v_emb, self.max_free, self.eos, want_verifier_hidden=two_phase)
        sd = SpecDecodeEngine(self.v, sch.head)
        toks, st = sd.verify_free_tokens(pre["draft"], self.max_new, self.eos, drop_first=True,
                                         continue_ar=not two_phase, want_hidden=two_phase)
        extra = 0.0
        if two_phase:
            nh = sd.next_hidden if sd.next_hidden is not None else pre["verifier_hidden"]
            sd2 = SpecDecodeEngine(self.v, self.heads[cfg["decode"]], 1)
            toks, st2 = sd2.run(d_hidden, self.max_new, self.eos, cfg["mode"], nh)
            st = SDStats(st.total_drafted + st2.total_drafted, st.total_accepted + st2.total_accepted,
                         st.iterations + st2.iterations, st.fallback_steps + st2.fallback_steps,
                         st2.tokens, st.decode_ms + st2.decode_ms)
            extra = egpt_decode_ms      # the fused drafter consumes the drafter's full answer
        out = self._result(toks, st, pre["prefill_wall_ms"], extra,
                           {"egpt_ms": pre["egpt_prefill_ms"], "vl_ms": pre["vl_prefill_ms"],
                            "overlap_ms": max(0.0, pre["egpt_total_ms"] + pre["vl_prefill_ms"]
                                              - pre["prefill_wall_ms"])})
        out["num_free_tok"] = pre["n_free"]
        out["gap_ms"] = max(0.0, pre["vl_prefill_ms"] - pre["egpt_prefill_ms"])
        return out

    def _result(self, toks, st: SDStats, prefill_ms, extra_ms, pp) -> Dict:
        total = prefill_ms + extra_ms + st.decode_ms
        return {"tokens": toks, "parallel_prefill_ms": prefill_ms, "egpt_prefill_ms": pp.get("egpt_ms", 0.0),
                "vl_prefill_ms": pp.get("vl_ms", 0.0), "overlap_ms": pp.get("overlap_ms", 0.0),
                "egpt_decode_ms": extra_ms, "sd_decode_ms": st.decode_ms, "sd_total_ms": total,
                "accept_rate": st.accept_rate, "drafted": st.total_drafted, "accepted": st.total_accepted,
                "iterations": st.iterations, "fallback_steps": st.fallback_steps, "num_tokens": st.tokens}


# ------------------------------------------------------------------ aggregation
def _mean(rows: List[Dict], key: str) -> float:
    vals = [r[key] for r in rows if key in r]
    return float(np.mean(vals)) if vals else 0.0


def _std(rows: List[Dict], key: str) -> float:
    vals = [r[key] for r in rows if key in r]
    return float(np.std(vals)) if vals else 0.0


def aggregate(samples: List[Dict], configs: List[Dict]) -> Dict[str, Dict]:
    base = [s["baseline"] for s in samples]
    agg = {"VL baseline": {"speedup_mean": 1.0, "prefill_ms_mean": _mean(base, "prefill_ms"),
                           "prefill_ms_std": _std(base, "prefill_ms"), "decode_ms_mean": _mean(base, "decode_ms"),
                           "decode_ms_std": _std(base, "decode_ms"), "total_ms_mean": _mean(base, "total_ms"),
                           "total_ms_std": _std(base, "total_ms"), "accept_rate_mean": 0.0,
                           "n_samples": len(base)}}
    for cfg in configs:
        rows = [s["configs"][cfg["name"]] for s in samples if cfg["name"] in s["configs"]]
        if not rows:
            continue
        e = {"speedup_mean": _mean(rows, "speedup"), "speedup_std": _std(rows, "speedup"),
             "prefill_ms_mean": _mean(rows, "parallel_prefill_ms"), "decode_ms_mean": _mean(rows, "sd_decode_ms"),
             "total_ms_mean": _mean(rows, "sd_total_ms"), "accept_rate_mean": _mean(rows, "accept_rate"),
             "overlap_ms_mean": _mean(rows, "overlap_ms"), "n_samples": len(rows),
             "lossless_all": all(r["lossless"] for r in rows),
             "exact_all": all(r.get("exact", r["lossless"]) for r in rows)}
        if any("num_free_tok" in r for r in rows):
            e["num_free_tok_mean"] = _mean(rows, "num_free_tok")
        agg[cfg["name"]] = e
    return agg


def stage_summary(rows: List[Dict]) -> Dict[str, float]:
    out = {k + "_mean": _mean(rows, k) for k in ("vision_ms", "prefill_ms", "decode_ms", "total_ms", "num_tokens",
                                                   "prefill_len")}
    out.update({k + "_std": _std(rows, k) for k in ("vision_ms", "prefill_ms", "decode_ms")})
    d, n = out["decode_ms_mean"], out["num_tokens_mean"]
    out["decode_tok_per_s"] = 1e3 * n / d if d > 0 else 0.0
    return out


def write_report(result: Dict, out_dir: str, ts: str) -> Dict[str, str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = {"json": write_json(result, os.path.join(out_dir, f"e2e_wallclock_{ts}.json"))}
    agg = result["aggregate"]
    rows = []
    for name, e in agg.items():
        rows.append((name, f"{e['prefill_ms_mean']:.1f}", f"{e['decode_ms_mean']:.1f}", f"{e['total_ms_mean']:.1f}",
                     "---" if name == "VL baseline" else f"{e['accept_rate_mean']:.1%}",
                     f"{e['speedup_mean']:.2f}x",
                     f"{e['num_free_tok_mean']:.1f}" if "num_free_tok_mean" in e else "---",
                     "" if name == "VL baseline" else ("yes" if e.get("exact_all", e.get("lossless_all")) else
                                                     ("near-tie" if e.get("lossless_all") else "NO"))))
    md = [f"# End-to-end wall-clock speculative decoding ({ts})", "",
          f"Samples: {result['n_measured']} measured (+{result['config']['warmup']} warm-up), "
          f"max_new_tokens {result['config']['max_new_tokens']}, gamma {result['config']['gamma']}, "
          f"data: {result['config']['data']}", "",
          markdown_table(["config", "prefill ms", "decode ms", "total ms", "accept", "speedup", "free tok",
                          "lossless"], rows), ""]
    st = result.get("stage_timing")
    if st:
        md += ["## 3-stage timing", "",
               markdown_table(["model", "vision ms", "prefill ms", "decode ms", "decode tok/s", "prefill tokens"],
                              [(k, f"{v['vision_ms_mean']:.2f}", f"{v['prefill_ms_mean']:.2f}",
                                f"{v['decode_ms_mean']:.2f}", f"{v['decode_tok_per_s']:.1f}",
                                f"{v['prefill_len_mean']:.0f}") for k, v in st.items()]), ""]
    paths["md"] = os.path.join(out_dir, f"e2e_wallclock_{ts}.md")
    with open(paths["md"], "w") as f:
        f.write("\n".join(md))
    paths.update(generate_graphs(agg, st, out_dir, ts))
    return paths


def generate_graphs(agg: Dict, stage_timing: Optional[Dict], out_dir: str, ts: str) -> Dict[str, str]:
    """speedup bars, accept-rate bars, prefill/decode stacks, stage timing."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        return {}
    names = [n for n in agg if n != "VL baseline"]
    out = {}
    if names:
        fig, ax = plt.subplots(figsize=(max(6, 0.7 * len(names) + 3), 4))
        vals = [agg[n]["speedup_mean"] for n in names]
        bars = ax.bar(names, vals, color=["seagreen" if v >= 1 else "indianred" for v in vals])
        ax.axhline(1.0, color="gray", ls="--")
        for b, v in zip(bars, vals):
            ax.text(b.get_x() + b.get_width() / 2, v, f"{v:.2f}x", ha="center", va="bottom", fontsize=8)
        ax.set_ylabel("speedup vs verifier baseline")
        ax.tick_params(axis="x", rotation=45)
        fig.tight_layout()
        out["speedup_png"] = os.path.join(out_dir, f"e2e_speedup_{ts}.png")
        fig.savefig(out["speedup_png"], dpi=120)
        plt.close(fig)
        fig, ax = plt.subplots(figsize=(max(6, 0.7 * len(names) + 3), 4))
        ax.bar(names, [agg[n]["accept_rate_mean"] for n in names], color="steelblue")
        ax.set_ylim(0, 1.05)
        ax.set_ylabel("draft acceptance rate")
        ax.tick_params(axis="x", rotation=45)
        fig.tight_layout()
        out["accept_png"] = os.path.join(out_dir, f"e2e_accept_{ts}.png")
        fig.savefig(out["accept_png"], dpi=120)
        plt.close(fig)
    allnames = list(agg)
    fig, ax = plt.subplots(figsize=(max(6, 0.7 * len(allnames) + 3), 4))
    pre = [agg[n]["prefill_ms_mean"] for n in allnames]
    dec = [agg[n]["total_ms_mean"] - agg[n]["prefill_ms_mean"] for n in allnames]
    ax.bar(allnames, pre, label="prefill", color="tab:blue")
    ax.bar(allnames, dec, bottom=pre, label="decode (+drafter)", color="tab:orange")
    ax.set_ylabel("ms")
    ax.legend()
    ax.tick_params(axis="x", rotation=45)
    fig.tight_layout()
    out["breakdown_png"] = os.path.join(out_dir, f"e2e_breakdown_{ts}.png")
    fig.savefig(out["breakdown_png"], dpi=120)
    plt.close(fig)
    if stage_timing:
        fig, ax = plt.subplots(figsize=(6, 4))
        models = list(stage_timing)
        bottom = np.zeros(len(models))
        for key, c in (("vision_ms_mean", "tab:purple"), ("prefill_ms_mean", "tab:blue"),
                       ("decode_ms_mean", "tab:green")):
            vals = np.array([stage_timing[m][key] for m in models])
            ax.bar(models, vals, bottom=bottom, label=key.replace("_ms_mean", ""), color=c)
            bottom += vals
        ax.set_ylabel("ms")
        ax.legend()
        fig.tight_layout()
        out["stages_png"] = os.path.join(out_dir, f"e2e_stages_{ts}.png")
        fig.savefig(out["stages_png"], dpi=120)
        plt.close(fig)
    return out


# -------------------------------------------------------------------------- CLI
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="E2E wall-clock SD benchmark (all methods and baselines)")
    p.add_argument("--dataset_dir", default=None, help="instruction dataset; synthetic if omitted")
    p.add_argument("--adapter_dir", default=None)
    p.add_argument("--max_samples", type=int, default=50)
    p.add_argument("--max_new_tokens", type=int, default=50)
    p.add_argument("--gamma", type=int, default=5)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--questions_file", default=None)
    p.add_argument("--max_questions", type=int, default=10)
    p.add_argument("--output_dir", default=None)
    p.add_argument("--configs", default=None, help="comma-separated config names (default: all)")
    p.add_argument("--event_image_key", default="event_image")
    p.add_argument("--egpt_model_path", default=None)
    p.add_argument("--vl_model_path", default=None)
    from ..quant import add_quant_args
    add_quant_args(p)
    p.add_argument("--random_init", default=None, choices=["7b", "small"],
                   help="random weights (and untrained adapters when --adapter_dir is absent)")
    p.add_argument("--stage_samples", type=int, default=20)
    return p


def _random_adapters(hidden: int, small: bool) -> Dict[str, torch.nn.Module]:
    from ..align.adapters import create_adapter
    kw = dict(num_heads=4, ffn_dim=256, bottleneck_dim=64) if small else {}
    torch.manual_seed(0)
    out = {"L1": create_adapter(1, hidden, **({"bottleneck_dim": 64} if small else {})),
           "L4": create_adapter(4, hidden, **({k: v for k, v in kw.items() if k != "bottleneck_dim"})),
           "B1": create_adapter(5, hidden, **({k: v for k, v in kw.items() if k != "bottleneck_dim"})),
           "L5F": create_adapter(6, hidden, **({k: v for k, v in kw.items() if k != "bottleneck_dim"}))}
    return out


def load_everything(args):
    from ..data.tokenizer import ByteTokenizer, load_tokenizer
    from ..models import ClipVisionConfig, EventChatConfig, EventChatModel
    from ..models.video_llava import VideoLlavaConfig, VideoLlavaModel, VideoProcessor
    dev, bf = "cuda", torch.bfloat16
    if args.random_init:
        tok = ByteTokenizer()
        if args.random_init == "7b":
            ecfg, vcfg = EventChatConfig.eventgpt_7b(), VideoLlavaConfig.video_llava_7b()
        else:
            vis = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2)
            ecfg = EventChatConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=2,
                                   num_attention_heads=4, max_position_embeddings=4096,
                                   mm_visual_tower="small-vit", event_feature_adaptor=True)
            ecfg.vision_config = ClipVisionConfig(image_size=56, patch_size=14, **vis)
            text = EventChatConfig(vocab_size=512, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
                                   num_attention_heads=4, max_position_embeddings=4096)
            vcfg = VideoLlavaConfig(text, ClipVisionConfig(image_size=56, patch_size=14, **vis), 4, -2, 510, 511)
        egpt = EventChatModel.from_config(ecfg, bf, dev, seed=0)
        vl = VideoLlavaModel.from_config(vcfg, bf, dev, seed=1)
        vproc = VideoProcessor(vcfg, tok)
        etok = tok
    else:
        egpt = EventChatModel.from_pretrained(args.egpt_model_path, bf, device=dev)
        vl = VideoLlavaModel.from_pretrained(args.vl_model_path, bf, device=dev)
        etok = load_tokenizer(args.egpt_model_path)
        vproc = VideoProcessor(vl.config, load_tokenizer(args.vl_model_path))
    from ..quant import apply_quant_args
    apply_quant_args(args, egpt, vl)
    ckpts = find_adapter_checkpoints(args.adapter_dir)
    if ckpts:
        adapters = {lv: load_any_adapter(p, dev)[0] for lv, p in ckpts.items()}
    elif args.random_init:
        adapters = _random_adapters(vl.config.text_config.hidden_size, args.random_init == "small")
    else:
        adapters = {}
    return egpt, etok, vl, vproc, adapters


def main(argv=None) -> Dict:
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("the end-to-end benchmark runs on the serving engines and needs a GPU")
    ts = datetime.now().strftime("%Y%m%d_%H%M%S")
    out_dir = args.output_dir or os.path.join(os.getcwd(), "tasks")
    data_kind = "real"
    if args.dataset_dir is None:
        import tempfile

        from ..pipeline.dataset import make_synthetic_dataset
        args.dataset_dir = make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_e2e_"),
                                                  n_samples=args.max_samples + args.warmup,
                                                  n_events=60000, width=640, height=480)
        data_kind = "synthetic"
    ds = InstructionDataset(args.dataset_dir, args.max_samples + args.warmup)
    questions = load_questions(args.questions_file or os.path.join(args.dataset_dir, "top50_questions.json"),
                               args.max_questions)
    egpt, etok, vl, vproc, adapters = load_everything(args)
    bench = E2EBenchmark(egpt, etok, vl, vproc, adapters, args.gamma, args.max_new_tokens, args.event_image_key)
    configs = build_sd_configs(list(adapters), args.configs.split(",") if args.configs else None)
    print(f"[e2e] {len(ds)} scenes x {len(questions)} questions, configs: {[c['name'] for c in configs]}", flush=True)

    # 3-stage timing pass
    stage = {"EventGPT": [], "Video-LLaVA": []}
    for i, sample in enumerate(ds.samples[: args.warmup + args.stage_samples]):
        inp = bench.prep.load(ds, sample, questions[0])
        if inp is None:
            continue
        e = time_3stage("egpt", bench.prep, inp, args.max_new_tokens, bench.eos)
        v = time_3stage("vl", bench.prep, inp, args.max_new_tokens, bench.eos)
        if i >= args.warmup:
            stage["EventGPT"].append(e)
            stage["Video-LLaVA"].append(v)
    stage_timing = {k: stage_summary(v) for k, v in stage.items() if v}

    samples = []
    for si, sample in enumerate(ds.samples):
        for qi, q in enumerate(questions):
            inp = bench.prep.load(ds, sample, q)
            if inp is None:
                continue
            r = bench.run_sample(inp, configs)
            r.update(sample=si, question=qi)
            if si >= args.warmup:
                samples.append(r)
        print(f"[e2e] scene {si + 1}/{len(ds)} done", flush=True)
    result = {"timestamp": ts, "config": {"dataset_dir": args.dataset_dir, "max_samples": args.max_samples,
                                          "max_new_tokens": args.max_new_tokens, "gamma": args.gamma,
                                          "warmup": args.warmup, "data": data_kind,
                                          "weights": args.random_init or "checkpoint", "weight_format": args.quant_type if args.use_4bit else "bf16",
                                          "adapters": sorted(adapters)},
              "n_measured": len(samples), "stage_timing": stage_timing,
              "aggregate": aggregate(samples, configs), "samples": samples}
    paths = write_report(result, out_dir, ts)
    for name, e in result["aggregate"].items():
        print(f"[e2e] {name:12s} total {e['total_ms_mean']:8.1f} ms  speedup {e['speedup_mean']:.2f}x  "
              f"accept {e['accept_rate_mean']:.1%}", flush=True)
    print(f"[e2e] wrote {paths['json']}", flush=True)
    return result


if __name__ == "__main__":
    main()
