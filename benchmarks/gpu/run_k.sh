#!/bin/bash
# GPU session K: wide prefill attention: numerics, then prefill lengths with it on / off.
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
step 300 t_wide.log $PT tests/test_attn_wide_gpu.py || { grep -E "PASSED|FAILED|^E  " gpurun_out/t_wide.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/t_wide.log | tail -3
step 400 t_ops.log $PT tests/test_ops_gpu.py tests/test_train_kernels_gpu.py tests/test_engine_gpu.py -q || { grep -E "FAILED|^E  " gpurun_out/t_ops.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/t_ops.log | tail -3
step 300 b_prefill.log python benchmarks/bench_prefill.py 636 1024 2115 4643
tail -4 gpurun_out/b_prefill.log
EVGPT_ATTN_WIDE=0 step 300 b_prefill_narrow.log python benchmarks/bench_prefill.py 636 1024 2115 4643
tail -4 gpurun_out/b_prefill_narrow.log
EVGPT_ATTN_WIDE=2 step 300 b_prefill_forced.log python benchmarks/bench_prefill.py 636
tail -1 gpurun_out/b_prefill_forced.log
