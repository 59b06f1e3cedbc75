#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -4 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 600 t_attn.log python -u -m pytest tests/test_ops_gpu.py tests/test_attn_wide_gpu.py tests/test_train_kernels_gpu.py tests/test_engine_gpu.py tests/test_sd_gpu.py -x -q --timeout 300 --timeout-method thread
step 300 b_prefill.log python benchmarks/bench_prefill.py 636 2115 4643
step 300 b_vision.log python benchmarks/bench_vision.py
cd /tmp && export TMPDIR=/tmp
echo "== pmc"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_k -o k -- python3 $ROOT/benchmarks/bench_prefill.py 636 4643 > $ROOT/gpurun_out/pmc_k.log 2>&1 || exit 1
f=$(find $ROOT/gpurun_out/pmc_k -name '*counter_collection.csv' | head -1)
python3 $ROOT/benchmarks/pmc_summary.py "$f" > $ROOT/gpurun_out/pmc_k_summary.txt
rm -rf $ROOT/gpurun_out/pmc_k
grep attn $ROOT/gpurun_out/pmc_k_summary.txt | cut -c1-300
