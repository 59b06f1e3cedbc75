#!/bin/bash
# A/B of two builds of the extension on one box: bash benchmarks/gpu/ab_so.sh <other .so> <rounds> <script + args>
# runs the script alternately with the default build and with EVGPT_SO=<other>, printing both outputs.
set -u
other=$1; rounds=$2; shift 2
for r in $(seq 1 "$rounds"); do
  echo "== round $r: default build"; timeout -k 10 300 python -u "$@" 2>&1 | grep -v amdgpu.ids || exit 1
  echo "== round $r: $other"; EVGPT_SO=$PWD/$other timeout -k 10 300 python -u "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
