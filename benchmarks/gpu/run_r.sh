#!/bin/bash
# session R: kernel profiles of the fine-tuning step (full and projector)
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -3 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
cd /tmp && export TMPDIR=/tmp
for mode in full projector; do
step 400 prof_sft_$mode.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sft_$mode -o sft -- python3 $ROOT/benchmarks/bench_sft_step.py $mode 6
done
cd $ROOT
for mode in full projector; do
f=$(find gpurun_out/prof_sft_$mode -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 60 > gpurun_out/prof_sft_${mode}_summary.txt 2>&1
find gpurun_out/prof_sft_$mode -name '*kernel_trace.csv' -delete
done
head -40 gpurun_out/prof_sft_projector_summary.txt
