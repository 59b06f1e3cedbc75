#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 900 t_gemm.log python -u -m pytest tests/test_ops_gpu.py tests/test_train_kernels_gpu.py tests/test_adapter_kernels_gpu.py tests/test_gemm256_gpu.py tests/test_lora.py tests/test_engine_gpu.py tests/test_sd_gpu.py -x -q --timeout 300 --timeout-method thread
for mode in lora projector; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
done
step 300 b_default.log python bench.py --steps 64
