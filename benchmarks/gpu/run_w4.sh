#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_attn_wide_gpu.py tests/test_ops_gpu.py tests/test_train_kernels_gpu.py tests/test_engine_gpu.py -m gpu > gpurun_out/t_w4.log 2>&1 || { tail -40 gpurun_out/t_w4.log; exit 1; }
tail -2 gpurun_out/t_w4.log
for mode in 1 0; do
  echo "attn wide mode $mode"
  EVGPT_ATTN_WIDE=$mode timeout -k 10 200 python benchmarks/bench_prefill.py 636 836 || exit 1
  EVGPT_ATTN_WIDE=$mode timeout -k 10 200 python benchmarks/bench_vision.py 2>&1 | tail -3 || exit 1
done
