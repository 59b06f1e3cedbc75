#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gemm256_gpu.py tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_sd_gpu.py -m gpu > gpurun_out/t_glu.log 2>&1 || { grep -E "FAILED|^E  " gpurun_out/t_glu.log | tail -30; exit 1; }
tail -2 gpurun_out/t_glu.log
timeout -k 10 300 python benchmarks/bench_prefill.py 636 2115 4643 || exit 1
timeout -k 10 300 python bench.py --steps 64 --warmup 8 | tail -1 | cut -c1-400
