#!/bin/bash
# GPU session H: end-to-end effect of the 8-wave GEMM: prefill, SFT step, verifier prefill; all GPU tests.
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
step 1100 t_all.log python -u -m pytest -x -q --timeout 300 --timeout-method thread tests -m gpu || { grep -E "FAILED|^E  " gpurun_out/t_all.log | tail -40; exit 1; }
tail -3 gpurun_out/t_all.log
for mode in projector lora full; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
  tail -1 gpurun_out/sft_$mode.log
done
step 300 b_default.log python bench.py --steps 128 --warmup 16
tail -1 gpurun_out/b_default.log
step 300 b_prefill.log python benchmarks/bench_prefill.py
tail -3 gpurun_out/b_prefill.log
step 600 b_verify.log python benchmarks/bench_verify.py
tail -8 gpurun_out/b_verify.log
