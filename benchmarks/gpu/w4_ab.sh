#!/bin/bash
# The 4-bit projections on one box: per-launch times (rows kernel, then the generic kernel), the kernel's
# tests, and the NF4 decode bench at one, two and three streams with either.   bash benchmarks/gpu/w4_ab.sh
set -e
timeout -k 10 200 python -u benchmarks/bench_w4_gemv.py 2>&1 | grep -v amdgpu.ids
echo "== generic kernel (EVGPT_W4_ROW1=0)"
EVGPT_W4_ROW1=0 timeout -k 10 200 python -u benchmarks/bench_w4_gemv.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python -u -m pytest -q --timeout 120 --timeout-method thread tests/test_w4.py -m gpu 2>&1 | tail -2
for b in 1 2 3; do
  for v in 3 1; do
    echo "== streams $b, rows kernel up to $v rows"; env EVGPT_W4_ROWS_MAX=$v timeout -k 10 200 python -u bench.py --weight-format nf4 --batch $b --steps 20 --warmup 5 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  done
done
