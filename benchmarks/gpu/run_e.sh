#!/bin/bash
# GPU session E: voxel LDS kernel (tests + 1e7 bench), optimizer tests, engine test fix.
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
PT="python -u -m pytest -v --timeout 300 --timeout-method thread"
step 600 t_event.log $PT tests/test_event_ops_gpu.py tests/test_optim_gpu.py tests/test_golden_sample.py -m gpu
grep -E "FAILED|ERROR|passed|failed|^E  " gpurun_out/t_event.log | tail -30
step 600 t_engine.log $PT tests/test_engine_gpu.py -m gpu -k "greedy_graph or 7b or teacher"
grep -E "FAILED|ERROR|passed|failed|^E  " gpurun_out/t_engine.log | tail -30
step 300 b_voxel.log python benchmarks/bench_voxel.py 1e7
tail -4 gpurun_out/b_voxel.log
