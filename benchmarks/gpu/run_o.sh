#!/bin/bash
# GPU session O: optimizer norm kernel: tests + fine-tuning step times.
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
step 300 t_o.log $PT tests/test_optim_gpu.py tests/test_train_sft.py tests/test_lora.py -q -m gpu || { grep -E "PASSED|FAILED|^E  " gpurun_out/t_o.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/t_o.log | tail -3
for mode in projector lora full; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
  tail -1 gpurun_out/sft_$mode.log
done
