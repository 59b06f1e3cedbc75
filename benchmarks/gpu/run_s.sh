#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 900 t_attn.log python -u -m pytest tests/test_ops_gpu.py tests/test_attn_wide_gpu.py tests/test_train_kernels_gpu.py tests/test_engine_gpu.py tests/test_sd_gpu.py tests/test_adapter_kernels_gpu.py -x -q --timeout 300 --timeout-method thread
step 300 b_prefill.log python benchmarks/bench_prefill.py 636 4643
step 400 sft_projector.log python benchmarks/bench_sft_step.py projector 8
cd /tmp && export TMPDIR=/tmp
for what in prefill sft; do
  if [ $what = prefill ]; then prog="$ROOT/benchmarks/bench_prefill.py 636 4643"; else prog="$ROOT/benchmarks/bench_sft_step.py projector 4"; fi
  echo "== pmc $what"
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$what -o p -- python3 $prog > $ROOT/gpurun_out/pmc_$what.log 2>&1 || { tail -5 $ROOT/gpurun_out/pmc_$what.log; exit 1; }
  f=$(find $ROOT/gpurun_out/pmc_$what -name '*counter_collection.csv' | head -1)
  python3 $ROOT/benchmarks/pmc_summary.py "$f" > $ROOT/gpurun_out/pmc_${what}_sq_summary.txt
  rm -rf $ROOT/gpurun_out/pmc_$what
  grep attn $ROOT/gpurun_out/pmc_${what}_sq_summary.txt | sed 's/,.*derived/ .../; s/^\([^,]*\),.*\(mfma_busy.*\)$/\1  \2/'
done
