#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_ops_gpu.py tests/test_gemm256_gpu.py tests/test_engine_gpu.py -m gpu > gpurun_out/t_v.log 2>&1 || { tail -30 gpurun_out/t_v.log; exit 1; }
tail -2 gpurun_out/t_v.log
timeout -k 10 300 python benchmarks/scratch/bench_sk.py > gpurun_out/b_sk.log 2>&1 || { tail gpurun_out/b_sk.log; exit 1; }
cut -c1-60 gpurun_out/b_sk.log
timeout -k 10 300 python bench.py --steps 64 --warmup 8 > gpurun_out/b_default.log 2>&1 || exit 1
tail -1 gpurun_out/b_default.log
timeout -k 10 300 python benchmarks/bench_vision.py > gpurun_out/b_vision.log 2>&1 || exit 1
tail -2 gpurun_out/b_vision.log
for mode in projector full; do timeout -k 10 300 python benchmarks/bench_sft_step.py $mode 8 > gpurun_out/sft_$mode.log 2>&1 || exit 1; tail -1 gpurun_out/sft_$mode.log; done
