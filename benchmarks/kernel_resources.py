#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one HIP source (cross-compiles on a CPU host).

    python benchmarks/kernel_resources.py eventgpt_amd/csrc/kernels/gemv.hip [filter]

Runs hipcc with ``-Rpass-analysis=kernel-resource-usage`` under the package's own flags and
prints one line per kernel: demangled name, VGPRs (accumulators included), spills, LDS bytes
and waves per SIMD.  Used to check an edit against the occupancy the launch geometry needs.
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eventgpt_amd import build as B  # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    cmd = [B._hipcc(), *B.HIP_FLAGS, f"-I{B.CSRC}", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise SystemExit(1)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = [r_["name"] for r_ in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'spill':>5s} {'lds':>7s} {'occ':>4s} {'sgpr':>5s}")
    for r_, d in zip(rows, dem):
        d = re.sub(r"\(.*\)$", "", d).replace("void ", "")
        if flt and flt not in d:
            continue
        print(f"{d[:70]:70s} {r_.get('VGPRs', '?'):>5s} {r_.get('AGPRs', '?'):>5s} "
              f"{r_.get('VGPRs Spill', '?'):>5s} {r_.get('LDS Size [bytes/block]', '?'):>7s} "
              f"{r_.get('Occupancy [waves/SIMD]', '?'):>4s} {r_.get('SGPRs', '?'):>5s}")


if __name__ == "__main__":
    main()
