#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -8 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 200 cfg6.log python benchmarks/check_gemm_deep_ring.py
step 500 b_gemm6.log python benchmarks/bench_gemm.py
grep -v amdgpu gpurun_out/b_gemm6.log | cut -c1-150 | head -30
