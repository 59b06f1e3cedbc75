#!/bin/bash
# GPU session M: column split of partial rounds + attention MFMA order: tests, GEMM bench (8f), prefill.
set -u
mkdir -p gpurun_out
ROOT=$PWD
step() {  # step <limit_s> <log> <cmd...>
  local lim=$1 log=$2; shift 2
  echo "== $* (limit ${lim}s)" | tee -a $ROOT/gpurun_out/session.log
  timeout -k 10 "$lim" "$@" > "$ROOT/gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a $ROOT/gpurun_out/session.log
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "ABORT after $log (rc=$rc)" | tee -a $ROOT/gpurun_out/session.log; tail -30 "$ROOT/gpurun_out/$log"; exit $rc; fi
  return $rc
}
PT="python -u -m pytest -x -v --timeout 120 --timeout-method thread"
step 300 t_m.log $PT tests/test_gemm256_gpu.py tests/test_attn_wide_gpu.py tests/test_ops_gpu.py -q || { grep -E "PASSED|FAILED|^E  " gpurun_out/t_m.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/t_m.log | tail -3
step 300 b_gemm8f.log python benchmarks/bench_gemm.py 8f
tail -5 gpurun_out/b_gemm8f.log
step 300 b_prefill.log python benchmarks/bench_prefill.py 636 2115 4643
tail -3 gpurun_out/b_prefill.log
