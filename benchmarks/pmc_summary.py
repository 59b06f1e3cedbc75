"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: sums each counter over
dispatches and prints derived ratios.  ``python benchmarks/pmc_summary.py <csv> [<csv> ...]``"""
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main(paths):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", row.get("kernel_name", "?")))
                c = row.get("Counter_Name", row.get("counter_name"))
                v = float(row.get("Counter_Value", row.get("counter_value", 0)) or 0)
                agg[k][c] += v
                if c == sorted(agg[k])[0]:
                    calls[k] += 1
    counters = sorted({c for k in agg for c in agg[k]})
    print("kernel," + ",".join(counters) + ",derived")
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("FETCH_SIZE", 0))):
        a = agg[k]
        d = []
        if a.get("SQ_WAVE_CYCLES"):
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
                d.append(f"mfma_busy/wave_cycles={a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_WAVE_CYCLES'] / 4:.3f}")
            if "SQ_WAIT_ANY" in a:
                d.append(f"wait_any={a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.2f}")
        if a.get("SQ_LDS_IDX_ACTIVE"):
            d.append(f"lds_conflict={a.get('SQ_LDS_BANK_CONFLICT', 0) / a['SQ_LDS_IDX_ACTIVE']:.3f}")
        if a.get("TCC_HIT_sum") is not None and (a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0)) > 0:
            d.append(f"l2_hit={a['TCC_HIT_sum'] / (a['TCC_HIT_sum'] + a['TCC_MISS_sum']):.3f}")
        print(k + "," + ",".join(f"{a.get(c, 0):.4g}" for c in counters) + "," + " ".join(d))


if __name__ == "__main__":
    main(sys.argv[1:])
