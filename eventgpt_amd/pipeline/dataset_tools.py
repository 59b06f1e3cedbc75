"""Dataset construction and analysis tools.

* ``build_clips`` — cut DSEC-style recordings into sliding clips (50 % overlap) of a
  chosen duration, write ``event_npy/<seq>/<clip>.npy`` + the RGB frames that fall in
  the clip under ``video/<seq>/<clip>/`` + the instruction JSON
  [feasible/my_egpt_dsec_dataset/build_my_egpt_dsec_seq.py:75-330, generate_json.py].
* ``preprocess_event_images`` — event streams -> raster PNGs in ``event_image/``
  (5 slices), ``event_image_1f/`` (1 slice) or ``event_image_<N>Hz/``; skips existing
  outputs, updates the JSON key; CPU process pool or one GPU raster kernel per clip
  [preprocess_event_images.py:58-324].
* ``frames_to_video`` — clip frames -> ``mp4/<name>.mp4`` through ffmpeg when it is
  installed, else a ``video/<name>.npy`` frame stack the loaders read directly
  [preprocess_videos_to_mp4.py].
* ``generate_answers`` — pseudo-label every (clip, question) with a pluggable answerer
  and a confidence; resumable [generate_answers_qwen.py:113-383].
* ``analyze_dataset`` / ``top_questions`` — clip durations, event densities, question
  frequencies -> ``top50_questions.json`` [feasible/analysis_datasets/*.py].
* ``build_alignment_features`` — verifier vision features / prefill hidden / tokens per
  sample, resumable [feasible/my_dsec_align_dataset/build_my_dsec_align_dataset*.py].
"""
from __future__ import annotations

import argparse
import json
import os
import re
import shutil
import subprocess
from collections import Counter
from concurrent.futures import ProcessPoolExecutor, as_completed
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from ..data.dsec import DSECDirectory, EventFile
from ..data.events import event_frames, load_event_npy, save_event_npy
from .dataset import DEFAULT_QUESTION, JSON_NAME

CLIP_DURATIONS_US = (1_000_000, 5_000_000, 10_000_000)


# ---------------------------------------------------------------------- clips
def format_duration(duration_us: int) -> str:
    if duration_us % 1_000_000 == 0:
        return f"{duration_us // 1_000_000}s"
    if duration_us % 1000 == 0:
        return f"{duration_us // 1000}ms"
    return f"{duration_us}us"


def parse_duration_from_dir(path: str) -> Optional[float]:
    """``.../my_egpt_dsec_seq_500ms`` -> 0.5 (seconds)."""
    m = re.search(r"_(\d+)(ms|us|s)$", os.path.basename(os.path.normpath(str(path))))
    if not m:
        return None
    return int(m.group(1)) * {"s": 1.0, "ms": 1e-3, "us": 1e-6}[m.group(2)]


def sliding_clips(t_start: int, t_end: int, duration_us: int, overlap: float = 0.5) -> List[Tuple[int, int]]:
    """Complete windows ``[a, a + duration)`` inside ``[t_start, t_end]``, stepping by
    ``duration * (1 - overlap)``."""
    step = max(int(duration_us * (1 - overlap)), 1)
    out, a = [], t_start
    while a + duration_us <= t_end:
        out.append((a, a + duration_us))
        a += step
    return out


def calculate_potential_clips(sequence_duration_us: int, clip_durations_us: Iterable[int] = CLIP_DURATIONS_US):
    return {d: (0 if sequence_duration_us < d else max(1, len(sliding_clips(0, sequence_duration_us, d))))
            for d in clip_durations_us}


def build_clips(seq_dirs: Sequence[str], out_dir: str, clip_duration_us: int = 1_000_000,
                questions: Sequence[str] = (DEFAULT_QUESTION,), overlap: float = 0.5, copy_frames: bool = True,
                max_clips_per_seq: Optional[int] = None) -> List[Dict]:
    """Returns the instruction-JSON entries it wrote (one per clip, conversations left
    with the first question and an empty answer for the labeller to fill)."""
    entries: List[Dict] = []
    for seq in seq_dirs:
        d = DSECDirectory(seq)
        name = os.path.basename(os.path.normpath(str(seq)))
        if not d.events.event_file.exists():
            continue
        stamps = np.atleast_1d(d.images.timestamps) if (d.images.root / "timestamps.txt").exists() else None
        frames = d.images.image_files_rectified or d.images.image_files_distorted
        with EventFile(d.events.event_file) as ef:
            lo, hi = (int(stamps.min()), int(stamps.max())) if stamps is not None and stamps.size else ef.time_range
            if hi - lo < clip_duration_us:
                continue
            for k, (a, b) in enumerate(sliding_clips(lo, hi, clip_duration_us, overlap)):
                if max_clips_per_seq is not None and k >= max_clips_per_seq:
                    break
                ev = ef.by_timewindow(a, b)
                if ev["t"].size == 0:
                    continue
                clip = f"{k:06d}"
                rel = os.path.join(name, clip + ".npy")
                os.makedirs(os.path.join(out_dir, "event_npy", name), exist_ok=True)
                save_event_npy(os.path.join(out_dir, "event_npy", rel), ev)
                in_clip = [] if stamps is None else np.nonzero((stamps >= a) & (stamps < b))[0].tolist()
                if copy_frames and in_clip and frames:
                    vdir = os.path.join(out_dir, "video", name, clip)
                    os.makedirs(vdir, exist_ok=True)
                    for j in in_clip:
                        if j < len(frames):
                            shutil.copy2(frames[j], os.path.join(vdir, os.path.basename(str(frames[j]))))
                entries.append({"id": f"{name}_{clip}", "event_data": rel, "video_data": os.path.join(name, clip),
                                "time_start": a, "time_end": b, "num_events": int(ev["t"].size),
                                "image_indices": in_clip,
                                "conversations": [{"from": "human", "value": "<event>\n" + questions[0]},
                                                  {"from": "gpt", "value": ""}]})
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, JSON_NAME), "w") as f:
        json.dump(entries, f, indent=2)
    return entries


# --------------------------------------------------------- event -> PNG rasters
def _raster_key(num_frames: int, hz: Optional[int]) -> str:
    if hz is not None:
        return f"event_image_{hz}Hz"
    return "event_image_1f" if num_frames == 1 else "event_image"


def _frames_for(path: str, num_frames: int, conflict: str, device: Optional[str]) -> List[np.ndarray]:
    ev = load_event_npy(path, trusted=True)          # datasets built by this package
    if device:
        import torch

        from ..ops.event_ops import events_to_device, rasterize_gpu
        dev_ev = events_to_device(ev, torch.device(device))
        # one canvas extent for all slices of the clip (the whole stream's)
        return [im for im in rasterize_gpu(dev_ev, num_frames, None, conflict).cpu().numpy()]
    return event_frames(ev, num_frames, conflict=conflict)


def process_single_npy(args) -> Tuple[str, Optional[List[str]], bool]:
    """``(relative npy path, saved relative image paths | None, skipped?)``."""
    from PIL import Image
    npy_path, out_dir, rel, num_frames, conflict, device = args
    stem, parent = os.path.splitext(os.path.basename(rel))[0], os.path.dirname(rel)
    names = [os.path.join(parent, f"{stem}_{i}.png") for i in range(num_frames)]
    if all(os.path.exists(os.path.join(out_dir, n)) for n in names):
        return rel, names, True
    try:
        frames = _frames_for(npy_path, num_frames, conflict, device)
        os.makedirs(os.path.join(out_dir, parent), exist_ok=True)
        for n, im in zip(names, frames):
            tmp = os.path.join(out_dir, n) + ".tmp.png"
            Image.fromarray(np.ascontiguousarray(im)).save(tmp)
            os.replace(tmp, os.path.join(out_dir, n))
        return rel, names, False
    except Exception as e:           # one corrupt clip must not stop a dataset-wide run
        print(f"[preprocess] {npy_path}: {type(e).__name__}: {e}", flush=True)
        return rel, None, False


def find_json_file(data_dir: str) -> Optional[str]:
    import glob
    for pat in (JSON_NAME, "*_instruction*.json", "*Instruction*.json", "*.json"):
        for m in sorted(glob.glob(os.path.join(data_dir, pat))):
            if "output" not in os.path.basename(m).lower():
                return m
    return None


def preprocess_event_images(data_dir: str, num_workers: Optional[int] = None, num_frames: int = 5,
                            hz: Optional[int] = None, conflict: str = "negative",
                            device: Optional[str] = None) -> Dict[str, int]:
    """``conflict="negative"`` reproduces the reference's vectorised painter (positives
    first, negatives on top); ``"last"`` matches the inference-time loop painter."""
    if hz is not None:
        dur = parse_duration_from_dir(data_dir)
        if dur is None:
            raise ValueError(f"cannot read a clip duration from the directory name {data_dir!r}")
        num_frames = max(1, int(dur * hz))
    key = _raster_key(num_frames, hz)
    npy_root, out_root = os.path.join(data_dir, "event_npy"), os.path.join(data_dir, key)
    files = sorted(os.path.join(r, f) for r, _, fs in os.walk(npy_root) for f in fs if f.endswith(".npy"))
    if not files:
        return {"processed": 0, "skipped": 0, "failed": 0, "num_frames": num_frames}
    os.makedirs(out_root, exist_ok=True)
    jobs = [(p, out_root, os.path.relpath(p, npy_root), num_frames, conflict, device) for p in files]
    results: Dict[str, List[str]] = {}
    stats = {"processed": 0, "skipped": 0, "failed": 0, "num_frames": num_frames}

    def take(res):
        rel, names, skipped = res
        if names is None:
            stats["failed"] += 1
            return
        results[rel] = names
        stats["skipped" if skipped else "processed"] += 1
    workers = num_workers if num_workers is not None else max(1, min(16, (os.cpu_count() or 2) - 1))
    if device or workers <= 1:
        for j in jobs:
            take(process_single_npy(j))
    else:
        with ProcessPoolExecutor(max_workers=workers) as ex:
            for fut in as_completed([ex.submit(process_single_npy, j) for j in jobs]):
                take(fut.result())
    jpath = find_json_file(data_dir)
    if jpath:
        with open(jpath) as f:
            data = json.load(f)
        for s in data:
            if s.get("event_data") in results:
                s[key] = results[s["event_data"]]
    else:
        jpath = os.path.join(data_dir, JSON_NAME)
        data = [{"event_data": k, key: v} for k, v in sorted(results.items())]
    tmp = jpath + ".tmp"
    with open(tmp, "w") as f:
        json.dump(data, f, indent=2, ensure_ascii=False)
    os.replace(tmp, jpath)
    stats["json_key"] = key
    return stats


# ------------------------------------------------------------------ video files
def frames_to_video(frames_dir: str, out_base: str, fps: int = 20, prefer_mp4: bool = True) -> Optional[str]:
    """``out_base + '.mp4'`` via ffmpeg if available, else ``out_base + '.npy'`` ``[T,H,W,3]`` uint8."""
    names = sorted(n for n in os.listdir(frames_dir) if n.lower().endswith((".png", ".jpg", ".jpeg")))
    if not names:
        return None
    os.makedirs(os.path.dirname(os.path.abspath(out_base)), exist_ok=True)
    if prefer_mp4 and shutil.which("ffmpeg"):
        lst = out_base + ".txt"
        with open(lst, "w") as f:
            for n in names:
                f.write(f"file '{os.path.abspath(os.path.join(frames_dir, n))}'\nduration {1.0 / fps}\n")
        r = subprocess.run(["ffmpeg", "-y", "-loglevel", "error", "-f", "concat", "-safe", "0", "-i", lst,
                            "-vf", "pad=ceil(iw/2)*2:ceil(ih/2)*2", "-pix_fmt", "yuv420p", out_base + ".mp4"],
                           capture_output=True, text=True)
        os.remove(lst)
        if r.returncode == 0:
            return out_base + ".mp4"
    from ..data.events import load_image
    np.save(out_base + ".npy", np.stack([np.asarray(load_image(os.path.join(frames_dir, n))) for n in names]))
    return out_base + ".npy"


def preprocess_videos(data_dir: str, fps: int = 20) -> Dict[str, int]:
    """Every ``video/<seq>/<clip>/`` frame folder -> ``mp4/<seq>/<clip>.mp4`` (or ``video/<seq>/<clip>.npy``)."""
    root = os.path.join(data_dir, "video")
    n_mp4 = n_npy = 0
    for seq in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        for clip in sorted(os.listdir(os.path.join(root, seq))):
            src = os.path.join(root, seq, clip)
            if not os.path.isdir(src):
                continue
            mp4_base = os.path.join(data_dir, "mp4", seq, clip)
            if os.path.exists(mp4_base + ".mp4") or os.path.exists(src + ".npy"):
                continue
            out = frames_to_video(src, mp4_base if shutil.which("ffmpeg") else src, fps)
            n_mp4 += bool(out and out.endswith(".mp4"))
            n_npy += bool(out and out.endswith(".npy"))
    return {"mp4": n_mp4, "npy": n_npy}


# --------------------------------------------------------------- pseudo-labels
def generate_answers(data_dir: str, questions: Sequence[str], answer_fn: Callable[[Dict, str], Tuple[str, float]],
                     min_confidence: float = 0.0, max_samples: Optional[int] = None,
                     save_every: int = 50) -> Dict[str, int]:
    """Fill ``conversations`` of every sample with (question, answer) turns.
    ``answer_fn(sample, question) -> (text, confidence in [0,1])``; samples that already
    hold answers for all questions are skipped, the JSON is saved every ``save_every`` samples."""
    jpath = find_json_file(data_dir)
    with open(jpath) as f:
        data = json.load(f)
    done = skipped = low = 0

    def flush():
        with open(jpath + ".tmp", "w") as f:
            json.dump(data, f, indent=2, ensure_ascii=False)
        os.replace(jpath + ".tmp", jpath)
    for i, s in enumerate(data[:max_samples] if max_samples else data):
        have = {c["value"].replace("<event>\n", "") for c in s.get("conversations", []) if c["from"] == "human"}
        answered = [c for c in s.get("conversations", []) if c["from"] == "gpt" and c["value"]]
        if len(answered) >= len(questions) and all(q in have for q in questions):
            skipped += 1
            continue
        conv, confs = [], []
        for q in questions:
            text, conf = answer_fn(s, q)
            if conf < min_confidence:
                low += 1
                continue
            conv += [{"from": "human", "value": "<event>\n" + q}, {"from": "gpt", "value": text}]
            confs.append(float(conf))
        s["conversations"], s["answer_confidence"] = conv, confs
        done += 1
        if done % save_every == 0:
            flush()
    flush()
    return {"labelled": done, "skipped": skipped, "low_confidence": low}


def model_answerer(model, processor_or_tokenizer, dataset, kind: str = "video", max_new_tokens: int = 64):
    """An ``answer_fn`` backed by one of this package's models; confidence is the mean
    probability of the greedy tokens."""
    import torch

    from ..models.video_llava import VIDEO_PROMPT, load_video_frames

    @torch.no_grad()
    def answer(sample: Dict, question: str) -> Tuple[str, float]:
        if kind != "video":
            raise NotImplementedError("only the video verifier is wired as a labeller")
        src = dataset.video_source(sample)
        frames = load_video_frames(src, model.config.num_frames) if src else None
        if not frames:
            return "", 0.0
        inp = processor_or_tokenizer(VIDEO_PROMPT.format(query=question), frames)
        emb = model.prepare_embeds(inp["input_ids"].to(model.device),
                                   inp["pixel_values_videos"].to(model.device, model.dtype))
        lm = model.language_model
        out = lm.model(inputs_embeds=emb, use_cache=True)
        past, last, pos, toks, probs = out.past_key_values, out.last_hidden_state[:, -1], emb.shape[1], [], []
        eos = model.config.text_config.eos_token_id
        for _ in range(max_new_tokens):
            p = torch.softmax(lm.lm_head(last).float(), -1)
            conf, nxt = p.max(-1)
            toks.append(int(nxt)); probs.append(float(conf))
            if toks[-1] == eos:
                break
            out = lm.model(inputs_embeds=lm.model.embed_tokens(nxt)[:, None],
                           position_ids=torch.full((1, 1), pos, device=emb.device), past_key_values=past,
                           use_cache=True)
            past, last, pos = out.past_key_values, out.last_hidden_state[:, -1], pos + 1
        text = processor_or_tokenizer.tokenizer.decode(toks, skip_special_tokens=True).strip()
        return text, float(np.mean(probs)) if probs else 0.0
    return answer


# -------------------------------------------------------------------- analysis
def top_questions(samples: Sequence[Dict], k: int = 50) -> List[Dict]:
    c = Counter()
    for s in samples:
        for turn in s.get("conversations", []):
            if turn.get("from") == "human":
                c[turn["value"].replace("<event>", "").strip()] += 1
    return [{"question": q, "count": n} for q, n in c.most_common(k)]


def analyze_dataset(data_dir: str, max_samples: Optional[int] = None, write: bool = True) -> Dict:
    """Clip durations, event counts/densities, sensor sizes, question frequencies."""
    jpath = find_json_file(data_dir)
    with open(jpath) as f:
        samples = json.load(f)
    if max_samples:
        samples = samples[:max_samples]
    dur, cnt, sizes = [], [], Counter()
    for s in samples:
        p = os.path.join(data_dir, "event_npy", s.get("event_data") or "")
        if not os.path.isfile(p):
            continue
        ev = load_event_npy(p, trusted=True)
        t = ev["t"].astype(np.int64)
        dur.append(float(t.max() - t.min()) / 1e6 if t.size else 0.0)
        cnt.append(int(t.size))
        sizes[(int(ev["x"].max()) + 1, int(ev["y"].max()) + 1)] += 1
    d, c = np.array(dur or [0.0]), np.array(cnt or [0])
    rep = {"num_samples": len(samples), "num_event_files": len(dur),
           "duration_s": {"mean": float(d.mean()), "min": float(d.min()), "max": float(d.max())},
           "events": {"mean": float(c.mean()), "min": int(c.min()), "max": int(c.max())},
           "event_rate_per_s": float(c.sum() / d.sum()) if d.sum() > 0 else 0.0,
           "sensor_sizes": {f"{w}x{h}": n for (w, h), n in sizes.items()},
           "top_questions": top_questions(samples, 50)}
    if write:
        with open(os.path.join(data_dir, "top50_questions.json"), "w") as f:
            json.dump(rep["top_questions"], f, indent=2)
        with open(os.path.join(data_dir, "dataset_analysis.json"), "w") as f:
            json.dump(rep, f, indent=2)
    return rep


# ---------------------------------------------------- verifier feature dataset
def build_alignment_features(model, processor, dataset, out_dir: str, question: str = DEFAULT_QUESTION,
                             max_new_tokens: int = 20, resume: bool = True) -> Dict[str, int]:
    """Per sample: the verifier's projected video features (mean-pooled per frame), the
    prefill hidden state of the last prompt token and its greedy tokens, one ``.pt`` per
    sample plus ``metadata.json``; existing files are skipped when ``resume``."""
    import torch

    from ..models.video_llava import VIDEO_PROMPT, load_video_frames
    os.makedirs(out_dir, exist_ok=True)
    meta, done, skipped = [], 0, 0
    for i, s in enumerate(dataset.samples):
        path = os.path.join(out_dir, f"{i:06d}.pt")
        if resume and os.path.exists(path):
            skipped += 1
            meta.append({"index": i, "id": s.get("id"), "file": os.path.basename(path)})
            continue
        src = dataset.video_source(s)
        frames = load_video_frames(src, model.config.num_frames) if src else None
        if not frames:
            continue
        with torch.no_grad():
            inp = processor(VIDEO_PROMPT.format(query=question), frames)
            px = inp["pixel_values_videos"].to(model.device, model.dtype)
            feats = model.get_video_features(px)
            emb = model.merge_video_features(inp["input_ids"].to(model.device), feats)
            out = model.language_model.model(inputs_embeds=emb)
            toks = model.generate(inp["input_ids"].to(model.device), px, max_new_tokens=max_new_tokens,
                                  use_engine=False)[0]
        t = model.config.num_frames
        torch.save({"vision_features": feats[0].view(t, -1, feats.shape[-1]).mean(1).float().cpu(),
                    "prefill_hidden": out.last_hidden_state[0, -1].float().cpu(), "tokens": toks.cpu(),
                    "prefill_len": int(emb.shape[1])}, path + ".tmp")
        os.replace(path + ".tmp", path)
        meta.append({"index": i, "id": s.get("id"), "file": os.path.basename(path)})
        done += 1
        with open(os.path.join(out_dir, "metadata.json"), "w") as f:
            json.dump({"question": question, "samples": meta}, f, indent=2)
    return {"written": done, "skipped": skipped}


def main(argv=None):
    p = argparse.ArgumentParser(description="Dataset tools")
    sub = p.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("clips")
    a.add_argument("--dsec_root", required=True)
    a.add_argument("--output_dir", required=True)
    a.add_argument("--duration_us", type=int, default=1_000_000)
    a = sub.add_parser("event_images")
    a.add_argument("--data_dir", required=True)
    a.add_argument("--all_datasets", action="store_true")
    a.add_argument("--num_workers", type=int, default=None)
    a.add_argument("--num_frames", type=int, default=5)
    a.add_argument("--hz", type=int, default=None)
    a.add_argument("--device", default=None)
    a = sub.add_parser("videos")
    a.add_argument("--data_dir", required=True)
    a.add_argument("--fps", type=int, default=20)
    a = sub.add_parser("analyze")
    a.add_argument("--data_dir", required=True)
    args = p.parse_args(argv)
    if args.cmd == "clips":
        seqs = sorted(os.path.join(args.dsec_root, d) for d in os.listdir(args.dsec_root)
                      if os.path.isdir(os.path.join(args.dsec_root, d)))
        out = os.path.join(args.output_dir, f"my_egpt_dsec_seq_{format_duration(args.duration_us)}")
        res = {"clips": len(build_clips(seqs, out, args.duration_us)), "output": out}
    elif args.cmd == "event_images":
        dirs = [args.data_dir] if not args.all_datasets else sorted(
            os.path.join(args.data_dir, d) for d in os.listdir(args.data_dir) if d.startswith("my_egpt_dsec"))
        res = {d: preprocess_event_images(d, args.num_workers, args.num_frames, args.hz, device=args.device)
               for d in dirs}
    elif args.cmd == "videos":
        res = preprocess_videos(args.data_dir, args.fps)
    else:
        res = analyze_dataset(args.data_dir)
    print(json.dumps(res, indent=2, default=str))
    return res


if __name__ == "__main__":
    main()
