#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/pp $ROOT/gpurun_out/ph
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 600 pp.log python -u -m eventgpt_amd.benchmarks.parallel_prefill --random_init 7b --max_samples 4 --max_new_tokens 50 --output_dir $ROOT/gpurun_out/pp
step 600 ph.log python -u -m eventgpt_amd.benchmarks.prefill_hiding_bench --random_init 7b --max_samples 4 --max_new_tokens 50 --output_dir $ROOT/gpurun_out/ph
ls $ROOT/gpurun_out/pp $ROOT/gpurun_out/ph
