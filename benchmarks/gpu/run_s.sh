#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 400 t_ad.log python -u -m pytest tests/test_adapter_kernels_gpu.py tests/test_align.py -x -q --timeout 200 --timeout-method thread
for lv in 1 4; do
  step 400 ad_${lv}_intree.log python benchmarks/bench_adapter_train.py $lv 4096 64
  step 400 ad_${lv}_lib.log python benchmarks/bench_adapter_train.py $lv 4096 64 library
done
