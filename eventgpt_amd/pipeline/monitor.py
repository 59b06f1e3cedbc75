"""Progress monitor and job-chain driver for the extraction -> training -> evaluation
pipeline.

Counterpart of the reference's shell watchdogs (pipeline/adapter_train/
auto_train_pipeline.sh:28-56 polls a process pattern and ``index.json`` every five
minutes, then trains and evaluates level after level; monitor_extraction.sh /
monitor_progress.py print progress).  Here the extractor is identified by an exact
PID or a PID file (never by a command-line pattern), completion is "the store's
index holds the expected number of samples, or the extractor has exited", and every
stage is a subprocess whose exit code stops the chain.

    python -m eventgpt_amd.pipeline.monitor status STORE_DIR [--expected N]
    python -m eventgpt_amd.pipeline.monitor wait STORE_DIR --expected N [--pid P]
    python -m eventgpt_amd.pipeline.monitor auto_train --train_data S --test_data T \
           --levels 1,2,3,4 --expected_test 10000 [--lm_head vl_lm_head.pt]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def read_index(store_dir: str) -> Optional[Dict]:
    p = os.path.join(store_dir, "index.json")
    if not os.path.exists(p):
        return None
    try:
        with open(p) as f:
            return json.load(f)
    except (OSError, ValueError):          # caught mid-rewrite; the next poll sees it whole
        return None


def pid_alive(pid: Optional[int]) -> bool:
    if not pid or pid <= 0:
        return False
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


def extraction_progress(store_dir: str, expected: Optional[int] = None) -> Dict:
    """Samples and chunks written so far, the write rate over the chunks' mtimes, and
    the remaining time when the expected total is known."""
    idx = read_index(store_dir)
    if idx is None:
        return {"exists": False, "total_samples": 0, "n_chunks": 0, "expected": expected, "complete": False}
    chunks = idx.get("chunks", [])
    times = []
    for c in chunks:
        p = os.path.join(store_dir, c["path"])
        if os.path.exists(p):
            times.append((os.path.getmtime(p), c.get("n_samples", 0)))
    times.sort()
    rate = None
    if len(times) >= 2 and times[-1][0] > times[0][0]:
        rate = sum(n for _, n in times[1:]) / (times[-1][0] - times[0][0])
    total = int(idx.get("total_samples", 0))
    out = {"exists": True, "total_samples": total, "n_chunks": len(chunks),
           "chunk_size": idx.get("chunk_size"), "samples_per_s": rate, "expected": expected,
           "complete": bool(expected is not None and total >= expected),
           "last_write": time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(times[-1][0])) if times else None}
    if expected and rate:
        out["eta_s"] = max(expected - total, 0) / rate
    return out


def format_progress(p: Dict) -> str:
    if not p["exists"]:
        return "no index.json yet"
    s = f"{p['total_samples']}" + (f" / {p['expected']}" if p["expected"] else "") + f" samples in {p['n_chunks']} chunks"
    if p.get("samples_per_s"):
        s += f", {p['samples_per_s']:.2f} samples/s"
    if p.get("eta_s") is not None:
        s += f", ETA {p['eta_s'] / 3600:.1f} h"
    if p.get("last_write"):
        s += f", last write {p['last_write']}"
    return s


def wait_for_extraction(store_dir: str, expected: int, pid: Optional[int] = None, pid_file: Optional[str] = None,
                        poll_s: float = 300.0, timeout_s: Optional[float] = None, out=print,
                        sleep=time.sleep) -> Dict:
    """Block until the store holds ``expected`` samples.  Returns the final progress with
    ``status`` = "complete", "extractor_exited" (the watched process ended short of the
    target) or "timeout"."""
    t0 = time.time()
    if pid is None and pid_file and os.path.exists(pid_file):
        with open(pid_file) as f:
            pid = int(f.read().strip() or 0)
    watched = pid is not None
    while True:
        p = extraction_progress(store_dir, expected)
        out(f"[{time.strftime('%H:%M:%S')}] {format_progress(p)}")
        if p["complete"]:
            return dict(p, status="complete")
        if watched and not pid_alive(pid):
            p = extraction_progress(store_dir, expected)       # a last flush may just have landed
            return dict(p, status="complete" if p["complete"] else "extractor_exited")
        if timeout_s is not None and time.time() - t0 > timeout_s:
            return dict(p, status="timeout")
        sleep(poll_s)


def _run(cmd: Sequence[str], log_path: Optional[str], out=print) -> int:
    out(f"[{time.strftime('%H:%M:%S')}] $ {' '.join(cmd)}")
    if log_path:
        os.makedirs(os.path.dirname(os.path.abspath(log_path)), exist_ok=True)
        with open(log_path, "a") as log:
            return subprocess.run(list(cmd), stdout=log, stderr=subprocess.STDOUT).returncode
    return subprocess.run(list(cmd)).returncode


def latest_run_dir(task_root: str, level_name: str) -> Optional[str]:
    d = os.path.join(task_root, level_name)
    if not os.path.isdir(d):
        return None
    runs = sorted(r for r in os.listdir(d) if os.path.exists(os.path.join(d, r, "best_model.pt")))
    return os.path.join(d, runs[-1]) if runs else None


LEVEL_NAMES = {1: "L1", 2: "L2", 3: "L3", 4: "L4", 5: "L5", 6: "L5F"}


def auto_train(train_data: str, test_data: Optional[str], levels: Sequence[int], output_dir: str = "tasks",
               log_dir: Optional[str] = None, lm_head: Optional[str] = None, num_epochs: int = 50,
               batch_size: int = 64, extra_train_args: Sequence[str] = (), cpu: bool = False, out=print,
               runner=_run) -> List[Dict]:
    """Train (and, given test data, evaluate) one adapter level after the other; a failing
    stage stops the chain, like ``set -e`` in the reference's script."""
    results: List[Dict] = []
    for lv in levels:
        name = LEVEL_NAMES[lv]
        cmd = [sys.executable, "-m", "eventgpt_amd.align.trainer", "--train_data", train_data, "--adapter_level",
               str(lv), "--output_dir", output_dir, "--num_epochs", str(num_epochs), "--batch_size",
               str(batch_size), *extra_train_args]
        if cpu:
            cmd.append("--cpu")
        rc = runner(cmd, os.path.join(log_dir, f"train_{name}.log") if log_dir else None, out)
        rec = {"level": name, "train_rc": rc, "run_dir": latest_run_dir(output_dir, name)}
        results.append(rec)
        if rc != 0:
            out(f"training {name} failed (exit {rc}); stopping the chain")
            break
        if test_data and rec["run_dir"]:
            ev = [sys.executable, "-m", "eventgpt_amd.eval.acceptance", "--checkpoint",
                  os.path.join(rec["run_dir"], "best_model.pt"), "--test_data", test_data, "--output_dir",
                  os.path.join(rec["run_dir"], "eval"), "--no_plots"]
            if lm_head:
                ev += ["--lm_head", lm_head]
            if cpu:
                ev += ["--device", "cpu"]
            rec["eval_rc"] = runner(ev, os.path.join(log_dir, f"eval_{name}.log") if log_dir else None, out)
            if rec["eval_rc"] != 0:
                out(f"evaluation of {name} failed (exit {rec['eval_rc']}); stopping the chain")
                break
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="pipeline progress and job chaining")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("status")
    s.add_argument("store_dir")
    s.add_argument("--expected", type=int, default=None)
    w = sub.add_parser("wait")
    w.add_argument("store_dir")
    w.add_argument("--expected", type=int, required=True)
    w.add_argument("--pid", type=int, default=None)
    w.add_argument("--pid_file", default=None)
    w.add_argument("--poll_s", type=float, default=300.0)
    w.add_argument("--timeout_s", type=float, default=None)
    a = sub.add_parser("auto_train")
    a.add_argument("--train_data", required=True)
    a.add_argument("--test_data", default=None)
    a.add_argument("--expected_test", type=int, default=None, help="wait until the test store holds this many")
    a.add_argument("--pid", type=int, default=None)
    a.add_argument("--pid_file", default=None)
    a.add_argument("--poll_s", type=float, default=300.0)
    a.add_argument("--levels", default="1,2,3,4")
    a.add_argument("--output_dir", default="tasks")
    a.add_argument("--log_dir", default=None)
    a.add_argument("--lm_head", default=None)
    a.add_argument("--num_epochs", type=int, default=50)
    a.add_argument("--batch_size", type=int, default=64)
    a.add_argument("--cpu", action="store_true")
    args = ap.parse_args(argv)
    if args.cmd == "status":
        p = extraction_progress(args.store_dir, args.expected)
        print(format_progress(p))
        return p
    if args.cmd == "wait":
        r = wait_for_extraction(args.store_dir, args.expected, args.pid, args.pid_file, args.poll_s, args.timeout_s)
        print(f"status: {r['status']}")
        if r["status"] != "complete":
            raise SystemExit(1)
        return r
    if args.test_data and args.expected_test:
        r = wait_for_extraction(args.test_data, args.expected_test, args.pid, args.pid_file, args.poll_s)
        if r["status"] != "complete":
            raise SystemExit(f"test extraction did not finish ({r['status']})")
    res = auto_train(args.train_data, args.test_data, [int(x) for x in args.levels.split(",")], args.output_dir,
                     args.log_dir, args.lm_head, args.num_epochs, args.batch_size, cpu=args.cpu)
    bad = [r for r in res if r["train_rc"] != 0 or r.get("eval_rc", 0) != 0]
    if bad:
        raise SystemExit(1)
    return res


if __name__ == "__main__":
    main()
