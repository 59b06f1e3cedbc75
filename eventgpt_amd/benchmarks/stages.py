"""Stage-separated inference benchmarks (3/4/5-stage) for the drafter and the verifier.

Stages: S1 load (disk -> arrays), S2 preprocess (rasterise / resize+normalise /
tokenise), S3 vision tower + projector (+ adaptor, pooling), S4 LLM prefill, S5
greedy decode.  Reported per sample: stage times, TTFT, decode tok/s, prefill
length, KV-cache MB, output ids; aggregated: per-stage means and shares,
drafter-vs-verifier comparison, token acceptance between the two outputs,
drafter/verifier cost ratio ``c`` and the theoretical SD speedup
``E[accepted] / (c*gamma + 1)``; optional 1-frame-vs-8-frame comparison and a tensor
shape dump per stage.  Reference: feasible/benchmark_inference/benchmark_inference.py:63-80,
1192-1251; benchmark_inference_4stages.py:207-262; benchmark_inference_5stages.py:223-482,
562-786, 843-1142, 1144-1417; analyze_stage4_5_shapes.py.

On a GPU the LLM stages run on the graph-captured serving engine; elsewhere they run
on the module path, so the whole benchmark also works (slowly) on a CPU.
"""
from __future__ import annotations

import argparse
import os
import time
from datetime import datetime
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..constants import EVENT_TOKEN_INDEX
from ..data.conversation import prepare_event_prompt
from ..data.events import event_frames, load_event_npy, load_image
from ..data.tokenizer import tokenizer_event_token
from ..eval.reports import markdown_table, write_json
from ..eval.timers import StageTimer, estimate_kv_cache_mb, gpu_memory_mb
from ..models.video_llava import VIDEO_PROMPT, load_video_frames
from ..ops.event_ops import preprocess_frames
from ..pipeline.dataset import DEFAULT_QUESTION, InstructionDataset

STAGE_KEYS = ("stage1_time", "stage2_time", "stage3_time", "stage4_time", "stage5_time")
STAGE_NAMES = ("Load", "Preprocess", "Vision", "Prefill", "Decode")


# ------------------------------------------------------------ SD arithmetic
def calculate_acceptance_rate(draft_tokens: Sequence[int], target_tokens: Sequence[int]):
    """Position-wise match rate over the common length: ``(rate, matched, compared)``."""
    n = min(len(draft_tokens), len(target_tokens))
    if n == 0:
        return 0.0, 0, 0
    matched = sum(1 for i in range(n) if draft_tokens[i] == target_tokens[i])
    return matched / n, matched, n


def calculate_speculative_speedup(c: float, alpha: float, gamma: int) -> float:
    """``(1 - alpha^(gamma+1)) / (1 - alpha) / (c*gamma + 1)`` with ``c = T_draft / T_target``."""
    if c <= 0:
        return 0.0
    alpha = min(alpha, 0.99)
    expected = (1 - alpha ** (gamma + 1)) / (1 - alpha)
    return expected / (c * gamma + 1)


@torch.no_grad()
def compute_aligned_acceptance_rate(draft_tokens: Sequence[int], target_tokens: Sequence[int], token_adapter,
                                    device="cpu") -> Dict:
    """Acceptance after translating draft ids through a trained ``TokenAdapter``."""
    if token_adapter is None:
        return {"aligned_acceptance_rate": 0.0, "aligned_matched": 0, "aligned_total": 0, "method": "no_adapter"}
    if not draft_tokens or not target_tokens:
        return {"aligned_acceptance_rate": 0.0, "aligned_matched": 0,
                "aligned_total": max(len(draft_tokens), len(target_tokens)), "method": "length_mismatch"}
    logits = token_adapter(torch.tensor([list(draft_tokens)], device=device))[0]
    n = min(logits.shape[0], len(target_tokens))
    tgt = torch.tensor(list(target_tokens[:n]), device=logits.device)
    top1 = int((logits[:n].argmax(-1) == tgt).sum())
    top5 = int((logits[:n].topk(min(5, logits.shape[-1]), -1).indices == tgt[:, None]).any(-1).sum())
    return {"aligned_acceptance_rate": top1 / len(target_tokens), "aligned_matched": top1,
            "aligned_total": len(target_tokens), "aligned_top5_rate": top5 / len(target_tokens),
            "aligned_top5_matched": top5, "method": "token_adapter"}


# ---------------------------------------------------------------- LLM stages
def _use_engine(model, embeds) -> bool:
    return embeds.is_cuda and embeds.dtype == torch.bfloat16 and not isinstance(model, _ModulePath)


@torch.no_grad()
def llm_prefill_decode(model, embeds: torch.Tensor, max_new_tokens: int, eos: int, timer: StageTimer):
    """Stages 4 and 5 for prompt embeddings ``[S, H]``; returns ``(token ids, kv_mb)``."""
    lm = getattr(model, "language_model", model)
    if _use_engine(model, embeds):
        eng = model.engine()
        eng.eos = eos
        with timer.stage("stage4_time"):
            eng.reset()
            eng.prefill(embeds, select="greedy")
        with timer.stage("stage5_time"):
            toks = eng.generate_continue(max_new_tokens, eos)
        kv_mb = eng.kv_bytes_per_token * (embeds.shape[0] + toks.shape[0]) / 2 ** 20
        return toks.tolist(), kv_mb
    with timer.stage("stage4_time"):
        out = lm.model(inputs_embeds=embeds[None], use_cache=True)
        nxt = lm.lm_head(out.last_hidden_state[:, -1]).float().argmax(-1)
    toks, past, pos = [int(nxt)], out.past_key_values, embeds.shape[0]
    with timer.stage("stage5_time"):
        while len(toks) < max_new_tokens and toks[-1] != eos:
            out = lm.model(inputs_embeds=lm.model.embed_tokens(nxt)[:, None],
                           position_ids=torch.full((1, 1), pos, device=embeds.device),
                           past_key_values=past, use_cache=True)
            past, pos = out.past_key_values, pos + 1
            nxt = lm.lm_head(out.last_hidden_state[:, -1]).float().argmax(-1)
            toks.append(int(nxt))
    return toks, estimate_kv_cache_mb(past)


# --------------------------------------------------------------- per-model runs
class EventGPTStages:
    """Drafter pipeline from a raw event stream (``source="npy"``) or cached rasters."""

    def __init__(self, model, tokenizer, num_frames: int = 5, source: str = "image",
                 event_image_key: str = "event_image", max_new_tokens: int = 50):
        self.m, self.tok = model, tokenizer
        self.num_frames, self.source, self.key, self.max_new = num_frames, source, event_image_key, max_new_tokens

    def run(self, ds: InstructionDataset, sample: Dict, query: str) -> Optional[Dict]:
        m = self.m
        timer = StageTimer(m.device)
        with timer.stage("stage1_time", host=True):
            if self.source == "npy":
                path = ds.event_npy_path(sample)
                if path is None:
                    return None
                raw = load_event_npy(path, trusted=True)     # files of this package's own datasets
            else:
                paths = ds.event_image_paths(sample, self.key)[: self.num_frames]
                if not paths or not all(os.path.exists(p) for p in paths):
                    return None
                raw = [np.asarray(load_image(p)) for p in paths]
        with timer.stage("stage2_time", host=True):
            imgs = event_frames(raw, self.num_frames) if self.source == "npy" else raw
            px = list(preprocess_frames(imgs, m.device, m.get_visual_tower().event_processor.size, m.dtype))
            ids = tokenizer_event_token(prepare_event_prompt(query, "eventgpt_v1"), self.tok, EVENT_TOKEN_INDEX,
                                        return_tensors="pt")[None].to(m.device)
        with timer.stage("stage3_time"), torch.no_grad():
            feats = m.encode_event_frames(px)[None]
            emb = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, event_features=feats,
                                                         event_image_sizes=[list(imgs[0].shape[:2])],
                                                         already_adapted=True)[4][0]
        toks, kv_mb = llm_prefill_decode(m, emb, self.max_new, m.config.eos_token_id, timer)
        return _finish(timer, toks, emb.shape[0], kv_mb, self.tok,
                       shapes={"pixel_values": [len(px)] + list(px[0].shape), "event_features": list(feats.shape),
                               "inputs_embeds": list(emb.shape)})


def frames_to_grid(frames: Sequence, rows: int = 2, cols: int = 4) -> np.ndarray:
    """Tile ``rows*cols`` frames into one image: how a single-image verifier (LLaVA-1.5)
    is fed a clip (benchmark_inference.py:213-251).  Missing frames repeat the last one;
    all tiles take the first frame's size."""
    arrs = [np.asarray(f) for f in frames][: rows * cols]
    if not arrs:
        raise ValueError("no frames")
    while len(arrs) < rows * cols:
        arrs.append(arrs[-1])
    h, w = arrs[0].shape[:2]
    canvas = np.zeros((rows * h, cols * w) + arrs[0].shape[2:], dtype=arrs[0].dtype)
    for i, a in enumerate(arrs):
        if a.shape[:2] != (h, w):
            from PIL import Image
            a = np.asarray(Image.fromarray(a).resize((w, h)))
        r, c = divmod(i, cols)
        canvas[r * h:(r + 1) * h, c * w:(c + 1) * w] = a
    return canvas


class VideoLlavaStages:
    """Verifier pipeline.  ``grid=(rows, cols)`` tiles the sampled frames into ONE image
    and runs the same model on it (the LLaVA-1.5 way of seeing a clip: one image's worth
    of visual tokens instead of one per frame)."""

    def __init__(self, model, processor, max_new_tokens: int = 50, grid: Optional[Sequence[int]] = None):
        self.m, self.proc, self.max_new, self.grid = model, processor, max_new_tokens, grid

    def run(self, ds: InstructionDataset, sample: Dict, query: str) -> Optional[Dict]:
        m = self.m
        timer = StageTimer(m.device)
        with timer.stage("stage1_time", host=True):
            src = ds.video_source(sample)
            frames = load_video_frames(src, m.config.num_frames) if src else None
            if not frames:
                return None
        with timer.stage("stage2_time", host=True):
            if self.grid:
                frames = [frames_to_grid(frames, *self.grid)]
            inp = self.proc(VIDEO_PROMPT.format(query=query), frames, device=m.device, dtype=m.dtype)
            ids = inp["input_ids"].to(m.device)
            px = inp["pixel_values_videos"]
        with timer.stage("stage3_time"), torch.no_grad():
            feats = m.get_video_features(px)
            emb = m.merge_video_features(ids, feats)[0]
        toks, kv_mb = llm_prefill_decode(m, emb, self.max_new, m.config.text_config.eos_token_id, timer)
        return _finish(timer, toks, emb.shape[0], kv_mb, self.proc.tokenizer,
                       shapes={"pixel_values_videos": list(px.shape), "video_features": list(feats.shape),
                               "inputs_embeds": list(emb.shape)})


def _finish(timer: StageTimer, toks: List[int], prefill_len: int, kv_mb: float, tokenizer, shapes: Dict) -> Dict:
    r = {k: v / 1e3 for k, v in timer.result().items()}          # seconds, like the reference's records
    for k in STAGE_KEYS:
        r.setdefault(k, 0.0)
    r["total_time"] = sum(r[k] for k in STAGE_KEYS)
    r["ttft"] = r["total_time"] - r["stage5_time"]
    r["prefill_tokens"] = int(prefill_len)
    r["decode_tokens"] = len(toks)
    r["decode_tokens_per_sec"] = len(toks) / r["stage5_time"] if r["stage5_time"] > 0 else 0.0
    r["kv_cache_mb"] = kv_mb
    r["token_ids"] = list(toks)
    r["output"] = tokenizer.decode(toks, skip_special_tokens=True).strip() if tokenizer is not None else ""
    r["shapes"] = shapes
    return r


# ------------------------------------------------- instrumentation equivalence
@torch.no_grad()
def verify_decoupling(model, processor, frames: Sequence, query: str, max_new_tokens: int = 16) -> Dict:
    """The verifier's monolithic ``generate`` timed from the outside with forward hooks
    vs the explicit staged pipeline: both must emit the same tokens, and the hook-side
    stage times must add up to the monolithic wall time (the check the reference runs
    before trusting staged numbers, verify_videollava_decoupling.py:87-340).  Uses the
    module path on both sides so the hooks see every forward."""
    from ..eval.timers import PrefillDecodeHooks, VisionTimingHooks
    inp = processor(VIDEO_PROMPT.format(query=query), frames, device=model.device, dtype=model.dtype)
    ids, px = inp["input_ids"].to(model.device), inp["pixel_values_videos"]
    vis, pd = VisionTimingHooks(model), PrefillDecodeHooks(model)
    # the vision hook sits on the tower module itself
    tower = model.video_tower
    vis.hooks.append(tower.register_forward_pre_hook(lambda m, i: setattr(vis, "_a", vis._span())))
    vis.hooks.append(tower.register_forward_hook(lambda m, i, o: setattr(vis, "_b", vis._span())))
    pd.register_hooks()
    timer = StageTimer(model.device)
    try:
        with timer.stage("monolithic"):
            mono = model.generate(ids, px, max_new_tokens=max_new_tokens, use_engine=False)[0].tolist()
        pd.finalize_decode_time()
        hooked = {"vision_s": vis.vision_time, "prefill_s": pd.prefill_time, "decode_s": pd.decode_time,
                  "forwards": pd.forwards}
    finally:
        vis.unregister()
        pd.unregister_hooks()
    t2 = StageTimer(model.device)
    with t2.stage("stage3_time"):
        feats = model.get_video_features(px)
        emb = model.merge_video_features(ids, feats)[0]
    toks, _ = llm_prefill_decode(_ModulePath(model), emb, max_new_tokens, model.config.text_config.eos_token_id, t2)
    staged = {k: v / 1e3 for k, v in t2.result().items()}
    mono_s = timer.result()["monolithic"] / 1e3
    n = min(len(mono), len(toks))
    return {"identical_tokens": mono[:n] == toks[:n] and n > 0, "n_tokens": n, "monolithic_s": mono_s,
            "hooked": hooked, "staged": staged,
            "hook_coverage": (hooked["vision_s"] + hooked["prefill_s"] + hooked["decode_s"]) / mono_s if mono_s else 0.0}


class _ModulePath:
    """View of a model that makes ``llm_prefill_decode`` take the module path."""

    def __init__(self, model):
        self._m = model
        self.language_model = getattr(model, "language_model", model)

    def __getattr__(self, name):
        return getattr(self._m, name)

    def engine(self):
        raise RuntimeError("module path requested")


# ------------------------------------------------------------------ summaries
def summarize_5stage(results: List[Dict], model_name: str = "EventGPT", out=print) -> Dict:
    if not results:
        out(f"no results for {model_name}")
        return {}
    n = len(results)
    avg = {k: sum(r[k] for r in results) / n for k in STAGE_KEYS}
    total = sum(avg.values())
    s = {"samples": n, **{f"stage{i + 1}_avg": avg[k] for i, k in enumerate(STAGE_KEYS)},
         **{f"stage{i + 1}_std": float(np.std([r[k] for r in results])) for i, k in enumerate(STAGE_KEYS)},
         "total_avg": total, "ttft_avg": sum(r["ttft"] for r in results) / n,
         "prefill_tokens_avg": sum(r["prefill_tokens"] for r in results) / n,
         "decode_tokens_avg": sum(r["decode_tokens"] for r in results) / n,
         "decode_tps_avg": sum(r["decode_tokens_per_sec"] for r in results) / n,
         "kv_cache_mb_avg": sum(r["kv_cache_mb"] for r in results) / n,
         "decode_pct": 100 * avg["stage5_time"] / total if total else 0.0}
    s["prefill_pct"] = 100 - s["decode_pct"] if total else 0.0
    s["prefill_throughput"] = s["prefill_tokens_avg"] / avg["stage4_time"] if avg["stage4_time"] > 0 else 0.0
    out(f"--- 5-stage summary: {model_name} ({n} samples)")
    for i, (k, name) in enumerate(zip(STAGE_KEYS, STAGE_NAMES)):
        out(f"  S{i + 1} {name:<11} {avg[k] * 1e3:9.3f} ms ({100 * avg[k] / total if total else 0:5.1f}%)")
    out(f"  total {total * 1e3:.3f} ms, TTFT {s['ttft_avg'] * 1e3:.3f} ms, decode {s['decode_tps_avg']:.1f} tok/s, "
        f"prefill {s['prefill_tokens_avg']:.0f} tok ({s['prefill_throughput']:.0f} tok/s), "
        f"KV {s['kv_cache_mb_avg']:.1f} MB")
    return s


def compare_models(egpt: Dict, vl: Dict, gamma: int = 4, alpha: Optional[float] = None, out=print) -> Dict:
    if not egpt or not vl:
        return {}
    cmp = {}
    for label, key in (("load", "stage1_avg"), ("preprocess", "stage2_avg"), ("vision", "stage3_avg"),
                       ("prefill", "stage4_avg"), ("decode", "stage5_avg"), ("total", "total_avg"),
                       ("ttft", "ttft_avg")):
        cmp[f"{label}_speedup"] = vl[key] / egpt[key] if egpt[key] > 0 else 0.0
    d_tok = egpt["stage5_avg"] / max(egpt["decode_tokens_avg"], 1)
    t_tok = vl["stage5_avg"] / max(vl["decode_tokens_avg"], 1)
    cmp["c_ratio"] = d_tok / t_tok if t_tok > 0 else 0.0
    cmp["token_ratio"] = vl["prefill_tokens_avg"] / egpt["prefill_tokens_avg"] if egpt["prefill_tokens_avg"] else 0.0
    if alpha is not None:
        cmp["alpha"] = alpha
        cmp["theoretical_sd_speedup"] = calculate_speculative_speedup(cmp["c_ratio"], alpha, gamma)
    out(f"--- drafter vs verifier: total {cmp['total_speedup']:.2f}x, prefill {cmp['prefill_speedup']:.2f}x "
        f"({cmp['token_ratio']:.2f}x tokens), c = {cmp['c_ratio']:.3f}"
        + (f", alpha = {alpha:.3f}, SD speedup(gamma={gamma}) = {cmp['theoretical_sd_speedup']:.2f}x"
           if alpha is not None else ""))
    return cmp


def acceptance_over_samples(egpt_results: List[Dict], vl_results: List[Dict], token_adapter=None,
                            device="cpu") -> Dict:
    rates, aligned = [], []
    for e, v in zip(egpt_results, vl_results):
        rates.append(calculate_acceptance_rate(e["token_ids"], v["token_ids"])[0])
        if token_adapter is not None:
            aligned.append(compute_aligned_acceptance_rate(e["token_ids"], v["token_ids"], token_adapter,
                                                           device)["aligned_acceptance_rate"])
    out = {"acceptance_rate_mean": float(np.mean(rates)) if rates else 0.0, "n": len(rates)}
    if aligned:
        out["aligned_acceptance_rate_mean"] = float(np.mean(aligned))
    return out


def write_stage_report(result: Dict, out_dir: str, ts: str) -> Dict[str, str]:
    os.makedirs(out_dir, exist_ok=True)
    paths = {"json": write_json(result, os.path.join(out_dir, f"benchmark_5stages_{ts}.json"))}
    md = [f"# Stage benchmark ({ts})", "", f"data: {result['config']['data']}, weights: {result['config']['weights']}",
          ""]
    rows = []
    for name, s in result["summaries"].items():
        rows.append([name] + [f"{s[f'stage{i}_avg'] * 1e3:.3f}" for i in range(1, 6)]
                    + [f"{s['ttft_avg'] * 1e3:.3f}", f"{s['decode_tps_avg']:.1f}", f"{s['prefill_tokens_avg']:.0f}",
                       f"{s['kv_cache_mb_avg']:.1f}"])
    md += [markdown_table(["model", "S1 load ms", "S2 preprocess ms", "S3 vision ms", "S4 prefill ms",
                           "S5 decode ms", "TTFT ms", "decode tok/s", "prefill tok", "KV MB"], rows), ""]
    if result.get("comparison"):
        md += ["## Drafter vs verifier", "",
               markdown_table(["metric", "value"], [(k, f"{v:.4g}") for k, v in result["comparison"].items()]), ""]
    if result.get("shapes"):
        md += ["## Tensor shapes per stage", ""]
        for name, sh in result["shapes"].items():
            md += [f"* {name}: " + ", ".join(f"`{k}` {v}" for k, v in sh.items())]
        md.append("")
    paths["md"] = os.path.join(out_dir, f"benchmark_5stages_{ts}.md")
    with open(paths["md"], "w") as f:
        f.write("\n".join(md))
    return paths


# -------------------------------------------------------------------------- CLI
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="3/4/5-stage inference benchmark, drafter vs verifier")
    p.add_argument("--dataset_dir", default=None, help="instruction dataset; synthetic if omitted")
    p.add_argument("--eventgpt_model_path", default=None)
    p.add_argument("--videollava_model", default=None)
    from ..quant import add_quant_args
    add_quant_args(p)
    p.add_argument("--random_init", default=None, choices=["7b", "tiny"])
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    p.add_argument("--max_samples", type=int, default=None)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--max_new_tokens", type=int, default=50)
    p.add_argument("--eventgpt_only", action="store_true")
    p.add_argument("--compare_1vs8", action="store_true", help="drafter on 1 event frame vs verifier on 8 frames")
    p.add_argument("--from_npy", action="store_true", help="start stage 1 from the raw event stream")
    p.add_argument("--num_frames", type=int, default=5)
    p.add_argument("--verify_decoupling", action="store_true",
                   help="check hook-timed monolithic generate == staged pipeline on the first sample")
    p.add_argument("--llava15_grid", action="store_true",
                   help="also run the verifier on one 2x4 grid image of the frames (LLaVA-1.5 style)")
    p.add_argument("--speculative", action="store_true", help="report acceptance, c and theoretical SD speedup")
    p.add_argument("--gamma", type=int, default=4)
    p.add_argument("--use_token_adapter", action="store_true")
    p.add_argument("--token_adapter_path", default=None)
    p.add_argument("--output_json", default=None)
    p.add_argument("--output_dir", default=None)
    return p


def main(argv=None) -> Dict:
    from ..pipeline.extract_hidden_states import load_models
    args = build_parser().parse_args(argv)
    ts = datetime.now().strftime("%Y%m%d_%H%M%S")
    data_kind = "real"
    n_samples = (args.max_samples or 8) + args.warmup
    if args.dataset_dir is None:
        import tempfile

        from ..pipeline.dataset import make_synthetic_dataset
        big = args.random_init != "tiny"
        args.dataset_dir = make_synthetic_dataset(tempfile.mkdtemp(prefix="evgpt_stages_"), n_samples=n_samples,
                                                  n_events=130000 if big else 4000, width=640 if big else 160,
                                                  height=480 if big else 120)
        data_kind = "synthetic"
    ds = InstructionDataset(args.dataset_dir, n_samples if args.max_samples else -1)
    largs = argparse.Namespace(device=args.device, random_init=args.random_init,
                               egpt_model_path=args.eventgpt_model_path, vl_model_path=args.videollava_model,
                               use_4bit=args.use_4bit, quant_type=args.quant_type)
    egpt, tok, vl, vproc = load_models(largs)
    frames = 1 if args.compare_1vs8 else args.num_frames
    key = "event_image_1f" if args.compare_1vs8 else "event_image"
    runners = {"EventGPT": EventGPTStages(egpt, tok, frames, "npy" if args.from_npy else "image", key,
                                          args.max_new_tokens)}
    if not args.eventgpt_only:
        runners["Video-LLaVA"] = VideoLlavaStages(vl, vproc, args.max_new_tokens)
        if args.llava15_grid:
            runners["LLaVA-1.5 (2x4 grid)"] = VideoLlavaStages(vl, vproc, args.max_new_tokens, grid=(2, 4))
    records: Dict[str, List[Dict]] = {k: [] for k in runners}
    for i, sample in enumerate(ds.samples):
        q = ds.question(sample) or DEFAULT_QUESTION
        for name, r in runners.items():
            rec = r.run(ds, sample, q)
            if rec is not None and i >= args.warmup:
                rec["sample"] = i
                records[name].append(rec)
    decoupling = None
    if args.verify_decoupling and not args.eventgpt_only and len(ds.samples):
        src = ds.video_source(ds.samples[0])
        fr = load_video_frames(src, vl.config.num_frames) if src else None
        if fr:
            decoupling = verify_decoupling(vl, vproc, fr, ds.question(ds.samples[0]) or DEFAULT_QUESTION,
                                           min(args.max_new_tokens, 16))
            print(f"[stages] decoupling check: identical tokens = {decoupling['identical_tokens']}, hooks cover "
                  f"{100 * decoupling['hook_coverage']:.0f} % of the monolithic time")
    summaries = {k: summarize_5stage(v, k) for k, v in records.items() if v}
    result = {"timestamp": ts, "config": {"data": data_kind, "weights": args.random_init or "checkpoint", "weight_format": args.quant_type if args.use_4bit else "bf16",
                                          "num_frames": frames, "from_npy": args.from_npy,
                                          "max_new_tokens": args.max_new_tokens, "device": str(args.device)},
              "summaries": summaries, "memory": gpu_memory_mb(),
              "shapes": {k: v[0]["shapes"] for k, v in records.items() if v}, "samples": records}
    if decoupling is not None:
        result["decoupling_check"] = decoupling
    if "Video-LLaVA" in summaries and "EventGPT" in summaries:
        alpha = None
        if args.speculative:
            adapter = None
            if args.use_token_adapter and args.token_adapter_path:
                from ..align.token_align import TokenAdapter
                adapter = TokenAdapter.load(args.token_adapter_path, args.device)
            acc = acceptance_over_samples(records["EventGPT"], records["Video-LLaVA"], adapter, args.device)
            result["acceptance"] = acc
            alpha = acc.get("aligned_acceptance_rate_mean", acc["acceptance_rate_mean"])
        result["comparison"] = compare_models(summaries["EventGPT"], summaries["Video-LLaVA"], args.gamma, alpha)
    out_dir = args.output_dir or os.path.join(os.getcwd(), "tasks")
    paths = write_stage_report(result, out_dir, ts)
    if args.output_json:
        write_json(result, args.output_json)
    print(f"[stages] wrote {paths['json']}")
    return result


if __name__ == "__main__":
    main()
