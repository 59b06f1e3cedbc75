#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -2 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 400 t_train.log python -u -m pytest tests/test_train_kernels_gpu.py tests/test_optim_gpu.py -x -q --timeout 120 --timeout-method thread
for mode in full lora; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
done
