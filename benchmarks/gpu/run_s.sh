#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 400 t_lora.log python -u -m pytest tests/test_lora.py tests/test_train_sft.py tests/test_train_kernels_gpu.py -x -q --timeout 200 --timeout-method thread
step 400 sft_lora.log python benchmarks/bench_sft_step.py lora 8
cd /tmp && export TMPDIR=/tmp
echo "== prof lora"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_lora -o lora -- python3 $ROOT/benchmarks/bench_sft_step.py lora 6 > $ROOT/gpurun_out/prof_lora.log 2>&1 || { tail -5 $ROOT/gpurun_out/prof_lora.log; exit 1; }
cd $ROOT
f=$(find gpurun_out/prof_lora -name '*kernel_stats.csv' | head -1)
python -m eventgpt_amd.utils.profsum "$f" 40 > gpurun_out/prof_lora_summary.txt 2>&1
find gpurun_out/prof_lora -name '*kernel_trace.csv' -delete
echo "library GEMM kernels left:"; grep -c "Cijk" gpurun_out/prof_lora_summary.txt
head -14 gpurun_out/prof_lora_summary.txt
