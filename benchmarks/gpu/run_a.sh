#!/bin/bash
# GPU session A: comm / ops / engine tests, then decode bench for each GEMV build variant,
# then a kernel-stats profile of the default build.  Every GPU step has its own time limit;
# a crash, abort or time-out ends the session (only plain test failures let it go on).
set -u
mkdir -p gpurun_out
step() {  # step <limit_s> <log> <cmd...>
  local lim=$1 log=$2; shift 2
  echo "== $* (limit ${lim}s)" | tee -a gpurun_out/session.log
  timeout -k 10 "$lim" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/session.log
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "ABORT after $log (rc=$rc)" | tee -a gpurun_out/session.log; tail -30 "gpurun_out/$log"; exit $rc; fi
  return $rc
}
step 900 t_comm.log python -u -m pytest tests/test_comm_gpu.py -x -v --timeout 120 --timeout-method thread
tail -5 gpurun_out/t_comm.log
step 900 t_ops.log python -u -m pytest tests/test_ops_gpu.py tests/test_w4.py -x -q --timeout 120 --timeout-method thread
tail -3 gpurun_out/t_ops.log
step 900 t_engine.log python -u -m pytest tests/test_engine_gpu.py tests/test_sd_gpu.py -x -v --timeout 120 --timeout-method thread
tail -15 gpurun_out/t_engine.log
step 300 b_default.log python bench.py --steps 128 --warmup 16
tail -1 gpurun_out/b_default.log
for v in un4 un4p1 un6o3 un8; do
  EVGPT_SO=$PWD/eventgpt_amd/_C_$v.so step 300 b_$v.log python bench.py --steps 128 --warmup 16
  tail -1 gpurun_out/b_$v.log
done
for ks in "4096x4096:2" "4096x4096:1" "4096x11008:2" "12288x4096:4" "12288x4096:1"; do
  EVGPT_GEMV_KS=$ks step 300 b_ks_${ks//[:x]/_}.log python bench.py --steps 128 --warmup 16
  echo "KS $ks: $(tail -1 gpurun_out/b_ks_${ks//[:x]/_}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["event_prefill_ms"])')"
done
cd /tmp && export TMPDIR=/tmp
step 300 prof.log rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8
cd $GRAFT_REPO_ROOT
python -m eventgpt_amd.utils.profsum "$(find gpurun_out/prof_default -name '*kernel_stats.csv' | head -1)" 30 > gpurun_out/prof_default_summary.txt 2>&1 || true
find gpurun_out/prof_default -name '*kernel_trace.csv' -delete
head -20 gpurun_out/prof_default_summary.txt
