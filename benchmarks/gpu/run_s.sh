#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -6 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 600 t_tp.log python -u -m pytest tests/test_engine_gpu.py tests/test_bench_launcher.py -x -v --timeout 420 --timeout-method thread -m gpu -k "tensor_parallel or rehearsal"
