#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/e2e
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -4 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 1000 e2e.log python -u -m eventgpt_amd.benchmarks.e2e_wallclock --random_init 7b --max_samples 4 --warmup 1 --max_new_tokens 50 --gamma 5 --stage_samples 4 --output_dir $ROOT/gpurun_out/e2e
ls $ROOT/gpurun_out/e2e | head
f=$(ls $ROOT/gpurun_out/e2e/*.md | head -1); [ -n "$f" ] && head -24 "$f"
step 300 t_rel.log python -u -m pytest tests/test_train_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k transposed
