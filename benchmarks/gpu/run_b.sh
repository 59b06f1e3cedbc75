#!/bin/bash
# GPU session B: training-kernel tests, GEMV unroll diagnostic, SFT step timing + kernel profile.
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
step 900 t_train.log python -u -m pytest tests/test_train_kernels_gpu.py -x -v --timeout 120 --timeout-method thread
tail -25 gpurun_out/t_train.log
step 600 t_ops.log python -u -m pytest tests/test_ops_gpu.py tests/test_train_sft.py tests/test_lora.py -x -q --timeout 120 --timeout-method thread
tail -3 gpurun_out/t_ops.log
step 300 b_default.log python bench.py --steps 128 --warmup 16
tail -1 gpurun_out/b_default.log
for v in un1 un2; do
  if [ -f eventgpt_amd/_C_$v.so ]; then
    EVGPT_SO=$PWD/eventgpt_amd/_C_$v.so step 300 b_$v.log python bench.py --steps 128 --warmup 16
    tail -1 gpurun_out/b_$v.log
  fi
done
for mode in projector lora full; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
  tail -1 gpurun_out/sft_$mode.log
done
cd /tmp && export TMPDIR=/tmp
step 400 prof_sft.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sft -o sft -- python3 $ROOT/benchmarks/bench_sft_step.py full 6
step 300 prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o bench -- python3 $ROOT/bench.py --steps 64 --warmup 8
cd $ROOT
for d in prof_sft prof_bench; do
  f=$(find gpurun_out/$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 30 > gpurun_out/${d}_summary.txt 2>&1
  find gpurun_out/$d -name '*kernel_trace.csv' -delete
  head -24 gpurun_out/${d}_summary.txt
done
