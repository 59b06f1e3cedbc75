#!/bin/bash
# Full GPU check: every GPU test, smoke, headline bench, prefill lengths, fine-tuning step, kernel profiles.
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
tail -2 gpurun_out/t_all.log
step 300 smoke.log python __graft_entry__.py smoke
tail -1 gpurun_out/smoke.log
step 300 b_default.log python bench.py
tail -1 gpurun_out/b_default.log
step 300 b_prefill.log python benchmarks/bench_prefill.py 636 2115 4643
tail -3 gpurun_out/b_prefill.log
step 400 b_gemm.log python benchmarks/bench_gemm.py
step 300 b_gemm_bwd.log python benchmarks/bench_gemm_bwd.py
for mode in projector lora full; do
  step 400 sft_$mode.log python benchmarks/bench_sft_step.py $mode 8
  tail -1 gpurun_out/sft_$mode.log
done
cd /tmp && export TMPDIR=/tmp
step 400 prof_prefill.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_prefill -o prefill -- python3 $ROOT/benchmarks/bench_prefill.py 4643
step 400 prof_sft.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sft -o sft -- python3 $ROOT/benchmarks/bench_sft_step.py full 6
step 300 prof_bench.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o bench -- python3 $ROOT/bench.py --steps 64 --warmup 8
cd $ROOT
for d in prof_prefill prof_sft prof_bench; do
  f=$(find gpurun_out/$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 40 > gpurun_out/${d}_summary.txt 2>&1
  find gpurun_out/$d -name '*kernel_trace.csv' -delete
done
head -12 gpurun_out/prof_sft_summary.txt
