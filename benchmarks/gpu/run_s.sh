#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
: > $ROOT/gpurun_out/serving_sweep.txt
for cfg in "32 96 64 4" "32 96 64 8" "32 96 64 16" "32 96 64 32" "8 96 64 4" "8 96 64 8"; do
  echo "== bench_serving $cfg"
  timeout -k 10 300 python benchmarks/bench_serving.py $cfg 2>/dev/null | tail -1 | tee -a $ROOT/gpurun_out/serving_sweep.txt
  [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
done
