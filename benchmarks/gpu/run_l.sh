#!/bin/bash
# GPU session L: wide attention after the softmax diet: numerics, prefill lengths, kernel profile at 4643.
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
step 300 t_wide.log $PT tests/test_attn_wide_gpu.py tests/test_ops_gpu.py tests/test_train_kernels_gpu.py -q || { grep -E "PASSED|FAILED|^E  " gpurun_out/t_wide.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/t_wide.log | tail -3
step 300 b_prefill.log python benchmarks/bench_prefill.py 636 1024 2115 4643
tail -4 gpurun_out/b_prefill.log
cd /tmp && export TMPDIR=/tmp
step 400 prof_prefill.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_prefill -o prefill -- python3 $ROOT/benchmarks/bench_prefill.py 4643
cd $ROOT
f=$(find gpurun_out/prof_prefill -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 12 > gpurun_out/prof_prefill_summary.txt 2>&1
find gpurun_out/prof_prefill -name '*kernel_trace.csv' -delete
head -8 gpurun_out/prof_prefill_summary.txt
