#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -2 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
cd /tmp && export TMPDIR=/tmp
step 400 prof_b32.log rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_b32 -o b32 -- python3 $ROOT/bench.py --batch 32 --steps 32 --warmup 4
cd $ROOT
f=$(find gpurun_out/prof_b32 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python -m eventgpt_amd.utils.profsum "$f" 30 > gpurun_out/prof_b32_summary.txt 2>&1
find gpurun_out/prof_b32 -name '*kernel_trace.csv' -delete
cat gpurun_out/prof_b32_summary.txt
