"""Summarise a rocprofv3 ``*_kernel_stats.csv``: short kernel names, one line each.

    python -m eventgpt_amd.utils.profsum gpurun_out/prof/vm/123_kernel_stats.csv [top_n]
"""
import csv
import re
import sys


def short(name: str, width: int = 70) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]{0,40}>)?)", name)
    base = m.group(1) if m else name
    return base[:width]


def main(path: str, top: int = 25) -> None:
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'pct':>6s}")
    for r in rows[:top]:
        t = float(r["TotalDurationNs"])
        print(f"{short(r['Name']):70s} {int(r['Calls']):7d} {t / 1e6:9.3f} "
              f"{float(r['AverageNs']) / 1e3:8.2f} {float(r['MinNs']) / 1e3:8.2f} {100 * t / total:6.2f}")
    print(f"total {total / 1e6:.3f} ms over {len(rows)} kernels")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
