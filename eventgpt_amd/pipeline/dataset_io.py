"""Local dataset snapshots: load, validate, convert.

Counterparts of the reference's small dataset scripts:
  * feasible/egpt_dataset/load_from_snapshot.py:14-53 and load_dataset.py — read the
    instruction JSON of a downloaded snapshot, optionally as a ``datasets.Dataset``,
    and cache it in Arrow format;
  * feasible/egpt_dataset/load_nimagenet.py — fetch a hub snapshot (here: resolve an
    *existing* local snapshot; nothing is downloaded unless the caller opts in);
  * feasible/my_egpt_dsec_dataset/test_dataset.py:13-83 — structural validation of an
    instruction dataset (required keys, human/gpt turn pair, event file present,
    event arrays well-formed).

Event files are read with the array-only loader of ``data.events``; a file that
needs general unpickling is reported as an issue, not executed.
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, List, Optional

import numpy as np

from ..data.events import load_event_npy

SNAPSHOT_JSON_NAMES = ("dataset_info.json", "EventGPT_Instruction_Subset.json",
                       "my_egpt_dsec_instruction_subset.json")


def find_snapshot_json(data_dir: str, json_name: Optional[str] = None) -> str:
    names = (json_name,) if json_name else SNAPSHOT_JSON_NAMES
    for n in names:
        p = os.path.join(data_dir, n)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no instruction JSON ({', '.join(names)}) in {data_dir}")


def load_snapshot(data_dir: str, json_name: Optional[str] = None, as_hf: bool = False):
    """The snapshot's instruction entries as a list of dicts, or as a
    ``datasets.Dataset`` (``as_hf=True``; an Arrow copy under ``arrow_format/`` is
    preferred when one exists)."""
    if as_hf:
        from datasets import Dataset
        arrow = os.path.join(data_dir, "arrow_format")
        if os.path.exists(os.path.join(arrow, "state.json")):
            return Dataset.load_from_disk(arrow)
    with open(find_snapshot_json(data_dir, json_name), encoding="utf-8") as f:
        entries = json.load(f)
    if not as_hf:
        return entries
    from datasets import Dataset
    return Dataset.from_list(entries)


def save_arrow(data_dir: str, json_name: Optional[str] = None, overwrite: bool = False) -> str:
    """Write the Arrow copy that ``load_snapshot(as_hf=True)`` picks up."""
    out = os.path.join(data_dir, "arrow_format")
    if os.path.exists(os.path.join(out, "state.json")) and not overwrite:
        return out
    from datasets import Dataset
    with open(find_snapshot_json(data_dir, json_name), encoding="utf-8") as f:
        Dataset.from_list(json.load(f)).save_to_disk(out)
    return out


def resolve_hub_snapshot(repo_id: str, local_dir: Optional[str] = None, allow_download: bool = False) -> str:
    """Path of a dataset snapshot: ``local_dir`` if it is populated, else the local hub
    cache.  A download happens only with ``allow_download=True``."""
    if local_dir and os.path.isdir(local_dir) and os.listdir(local_dir):
        return local_dir
    from huggingface_hub import snapshot_download
    return snapshot_download(repo_id=repo_id, repo_type="dataset", local_dir=local_dir,
                             local_files_only=not allow_download, max_workers=1)


# ------------------------------------------------------------------ validation
def validate_entry(entry: Dict, event_dir: str) -> Optional[str]:
    """``None`` if the entry is well-formed, else what is wrong with it."""
    missing = [k for k in ("id", "event_data", "conversations") if k not in entry]
    if missing:
        return f"missing fields {missing}"
    conv = entry["conversations"]
    if not isinstance(conv, list) or len(conv) != 2:
        return f"expected 2 conversation turns, got {len(conv) if isinstance(conv, list) else type(conv).__name__}"
    if conv[0].get("from") != "human" or conv[1].get("from") != "gpt":
        return "turns are not human -> gpt"
    if not os.path.exists(os.path.join(event_dir, entry["event_data"])):
        return f"event file {entry['event_data']} not found"
    return None


def event_file_stats(path: str) -> Dict:
    ev = load_event_npy(path)
    t = np.asarray(ev["t"])
    n = int(t.shape[0])
    return {"num_events": n, "time_range_us": int(t.max() - t.min()) if n else 0,
            "width": int(np.asarray(ev["x"]).max()) + 1 if n else 0,
            "height": int(np.asarray(ev["y"]).max()) + 1 if n else 0,
            "positive_fraction": float((np.asarray(ev["p"]) > 0).mean()) if n else 0.0}


def validate_dataset(dataset_dir: str, json_name: Optional[str] = None, event_subdir: str = "event_npy",
                     n_event_checks: int = 5, out=print) -> Dict:
    """Structural check of every entry and a content check of the first
    ``n_event_checks`` event files (-1: all)."""
    entries = load_snapshot(dataset_dir, json_name)
    event_dir = os.path.join(dataset_dir, event_subdir)
    issues: List[Dict] = []
    stats: List[Dict] = []
    valid = 0
    for i, e in enumerate(entries):
        why = validate_entry(e, event_dir)
        if why is not None:
            issues.append({"index": i, "id": e.get("id"), "issue": why})
            continue
        if n_event_checks < 0 or len(stats) < n_event_checks:
            try:
                s = event_file_stats(os.path.join(event_dir, e["event_data"]))
            except Exception as ex:          # unreadable / refused / malformed event file
                issues.append({"index": i, "id": e.get("id"), "issue": f"event data: {ex}"})
                continue
            stats.append({"index": i, **s})
        valid += 1
    for it in issues[:20]:
        out(f"entry {it['index']}: {it['issue']}")
    out(f"validation: {valid}/{len(entries)} valid entries, {len(issues)} issues")
    return {"n_entries": len(entries), "n_valid": valid, "issues": issues, "event_stats": stats}


def show_samples(dataset_dir: str, n: int = 3, json_name: Optional[str] = None, out=print) -> None:
    for i, e in enumerate(load_snapshot(dataset_dir, json_name)[:n]):
        conv = e.get("conversations", [{}, {}])
        out(f"entry {i}: id={e.get('id')} event={e.get('event_data')}")
        out(f"  Q: {conv[0].get('value', '').replace('<event>', '').strip()}")
        out(f"  A: {conv[1].get('value', '') if len(conv) > 1 else ''}")


def main(argv=None):
    ap = argparse.ArgumentParser(description="local dataset snapshots: validate / show / to_arrow")
    sub = ap.add_subparsers(dest="cmd", required=True)
    v = sub.add_parser("validate")
    v.add_argument("dataset_dir")
    v.add_argument("--json_name", default=None)
    v.add_argument("--event_subdir", default="event_npy")
    v.add_argument("--n_event_checks", type=int, default=5)
    v.add_argument("--report", default=None)
    s = sub.add_parser("show")
    s.add_argument("dataset_dir")
    s.add_argument("-n", type=int, default=3)
    a = sub.add_parser("to_arrow")
    a.add_argument("dataset_dir")
    a.add_argument("--overwrite", action="store_true")
    args = ap.parse_args(argv)
    if args.cmd == "validate":
        rep = validate_dataset(args.dataset_dir, args.json_name, args.event_subdir, args.n_event_checks)
        if args.report:
            with open(args.report, "w") as f:
                json.dump(rep, f, indent=2)
        return rep
    if args.cmd == "show":
        return show_samples(args.dataset_dir, args.n)
    print(save_arrow(args.dataset_dir, overwrite=args.overwrite))


if __name__ == "__main__":
    main()
