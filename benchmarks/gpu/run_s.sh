#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
step() { local t=$1 log=$2; shift 2; echo "== $*" ; timeout -k 10 $t "$@" > $ROOT/gpurun_out/$log 2>&1; local rc=$?; tail -4 $ROOT/gpurun_out/$log; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
step 900 rehearsal2_7b.log python bench.py --gpus 2 --rehearsal --steps 4 --warmup 2
