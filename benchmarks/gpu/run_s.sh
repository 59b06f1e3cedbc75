#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -2 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
for mode in projector lora full; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
done
cat gpurun_out/sft_projector.log gpurun_out/sft_lora.log gpurun_out/sft_full.log | grep "ms per step" > gpurun_out/sft_step_times.txt
