#!/bin/bash
# GPU session C: GEMM with buffer-load staging (numerics, prefill, SFT step), adapter kernels,
# device-resident SD rounds, splice kernel, engine tests, profiles.
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
step 600 t_adapter.log python -u -m pytest -v --timeout 120 --timeout-method thread tests/test_adapter_kernels_gpu.py tests/test_splice_gpu.py -k 'not match_module and not d512'
tail -25 gpurun_out/t_adapter.log
step 900 t_engine.log $PT tests/test_engine_gpu.py tests/test_sd_gpu.py tests/test_w4.py tests/test_event_ops_gpu.py tests/test_optim_gpu.py -q -m gpu
tail -15 gpurun_out/t_engine.log
step 300 b_default.log python bench.py --steps 128 --warmup 16
tail -1 gpurun_out/b_default.log
step 400 b_gemm.log python benchmarks/bench_gemm.py
tail -30 gpurun_out/b_gemm.log
for mode in projector lora full; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
  tail -1 gpurun_out/sft_$mode.log
done
step 600 b_verify.log python benchmarks/bench_verify.py
tail -30 gpurun_out/b_verify.log
cd /tmp && export TMPDIR=/tmp
step 400 prof_sft.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sft -o sft -- python3 $ROOT/benchmarks/bench_sft_step.py full 6
cd $ROOT
for d in prof_sft; do
  f=$(find gpurun_out/$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 30 > gpurun_out/${d}_summary.txt 2>&1
  find gpurun_out/$d -name '*kernel_trace.csv' -delete
  head -24 gpurun_out/${d}_summary.txt
done
