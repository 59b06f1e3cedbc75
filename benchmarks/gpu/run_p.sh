#!/bin/bash
# GPU session P: all GPU tests, then the headline bench (decode, prefill, vision).
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
step 1100 t_all.log python -u -m pytest -q --timeout 300 --timeout-method thread tests -m gpu || { grep -E "FAILED|^E  " gpurun_out/t_all.log | tail -40; exit 1; }
tail -3 gpurun_out/t_all.log
step 300 b_default.log python bench.py --steps 128 --warmup 16
tail -1 gpurun_out/b_default.log
step 300 b_vision.log python benchmarks/bench_vision.py
tail -5 gpurun_out/b_vision.log
