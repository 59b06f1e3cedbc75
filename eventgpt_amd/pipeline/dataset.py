"""Instruction-dataset access shared by extractors, trainers and benchmarks.

Layout (reference: feasible/my_egpt_dsec_dataset/preprocess_event_images.py:225-309,
pipeline/feature_extraction/extract_hidden_states.py:179-183, 259-263, 432-453):

    <dir>/EventGPT_Instruction_Subset.json   [{id, event_data, video_data?,
                                               event_image?[5], event_image_1f?[1],
                                               conversations:[{from,value},...]}]
    <dir>/event_npy/<event_data>             pickled-dict or plain event arrays
    <dir>/event_image/<name>.png             5-slice rasters
    <dir>/event_image_1f/<name>.png          single-slice raster
    <dir>/mp4/<video_data>.mp4               RGB clip (PyAV needed)
    <dir>/video/<video_data>/*.png|.npy      the same clip as frames (no PyAV needed)

``make_synthetic_dataset`` writes a small dataset of this shape (synthetic event
streams, rasters made by this package's rasteriser, frame stacks as ``.npy``) for
tests and for benchmarks on machines without the real data.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

JSON_NAME = "EventGPT_Instruction_Subset.json"
DEFAULT_QUESTION = "What are the key elements in this scene?"


class InstructionDataset:
    def __init__(self, dataset_dir: str, max_samples: int = -1, json_name: str = JSON_NAME):
        self.dir = str(dataset_dir)
        self.json_path = os.path.join(self.dir, json_name)
        with open(self.json_path) as f:
            self.samples: List[Dict] = json.load(f)
        if max_samples and max_samples > 0:
            self.samples = self.samples[:max_samples]

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, i: int) -> Dict:
        return self.samples[i]

    def __iter__(self):
        return iter(self.samples)

    # -- path helpers ---------------------------------------------------------
    def event_npy_path(self, sample: Dict) -> Optional[str]:
        name = sample.get("event_data")
        if not name:
            return None
        p = os.path.join(self.dir, "event_npy", name)
        return p if os.path.exists(p) else None

    def event_image_paths(self, sample: Dict, key: str = "event_image") -> List[str]:
        return [os.path.join(self.dir, key, n) for n in sample.get(key) or []]

    def video_source(self, sample: Dict) -> Optional[str]:
        """Frame directory / frame stack / container of the sample's RGB clip, in
        the order that needs the fewest optional dependencies."""
        name = sample.get("video_data")
        if not name:
            return None
        for cand in (os.path.join(self.dir, "video", name), os.path.join(self.dir, "video", name + ".npy"),
                     os.path.join(self.dir, "mp4", name + ".mp4")):
            if os.path.exists(cand):
                return cand
        return None

    def question(self, sample: Dict) -> str:
        for turn in sample.get("conversations", []):
            if turn.get("from") == "human":
                return turn["value"].replace("<event>", "").strip()
        return DEFAULT_QUESTION

    def save(self, path: Optional[str] = None) -> None:
        tmp = (path or self.json_path) + ".tmp"
        with open(tmp, "w") as f:
            json.dump(self.samples, f, indent=2)
        os.replace(tmp, path or self.json_path)


def load_questions(questions_file: Optional[str], max_questions: int = 10) -> List[str]:
    """``top50_questions.json`` = ``[{question, count}]``; falls back to the default question."""
    if questions_file and os.path.exists(questions_file):
        with open(questions_file) as f:
            data = json.load(f)
        qs = [q["question"] if isinstance(q, dict) else str(q) for q in data]
        return qs[:max_questions] if max_questions > 0 else qs
    return [DEFAULT_QUESTION]


def shard_pairs(n_samples: int, n_questions: int, rank: int = 0, world: int = 1,
                done: Optional[set] = None) -> List[tuple]:
    """(sample, question) pairs of this rank, round-robin so ranks finish together."""
    pairs = [(s, q) for s in range(n_samples) for q in range(n_questions)]
    pairs = pairs[rank::world]
    return [p for p in pairs if not done or p not in done]


def make_synthetic_dataset(out_dir: str, n_samples: int = 4, n_events: int = 20000, width: int = 160,
                           height: int = 120, video_frames: int = 10, seed: int = 0,
                           questions: Sequence[str] = (DEFAULT_QUESTION, "What is moving?")) -> str:
    from PIL import Image

    from ..data.events import event_frames, save_event_npy, synthetic_events
    rng = np.random.default_rng(seed)
    for sub in ("event_npy", "event_image", "event_image_1f", "video"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    samples = []
    for i in range(n_samples):
        name = f"seq{i:03d}"
        ev = synthetic_events(n_events, width, height, seed=seed + i)
        save_event_npy(os.path.join(out_dir, "event_npy", name + ".npy"), ev)
        imgs5 = event_frames(ev, 5)
        names5 = []
        for j, im in enumerate(imgs5):
            fn = f"{name}_{j}.png"
            Image.fromarray(im).save(os.path.join(out_dir, "event_image", fn))
            names5.append(fn)
        fn1 = f"{name}_1f.png"
        Image.fromarray(event_frames(ev, 1)[0]).save(os.path.join(out_dir, "event_image_1f", fn1))
        clip = rng.integers(0, 255, (video_frames, height, width, 3), dtype=np.uint8)
        np.save(os.path.join(out_dir, "video", name + ".npy"), clip)
        samples.append({"id": name, "event_data": name + ".npy", "video_data": name, "event_image": names5,
                        "event_image_1f": [fn1],
                        "conversations": [{"from": "human", "value": "<event>\n" + questions[i % len(questions)]},
                                          {"from": "gpt", "value": "A synthetic scene."}]})
    with open(os.path.join(out_dir, JSON_NAME), "w") as f:
        json.dump(samples, f, indent=2)
    with open(os.path.join(out_dir, "top50_questions.json"), "w") as f:
        json.dump([{"question": q, "count": 10 - k} for k, q in enumerate(questions)], f)
    return out_dir
