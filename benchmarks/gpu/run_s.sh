#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 400 l6_off.log python benchmarks/bench_lora_distill.py 10 off
step 400 l6_on.log python benchmarks/bench_lora_distill.py 10 on
step 400 t_lora.log python -u -m pytest tests/test_lora.py -x -q --timeout 200 --timeout-method thread
