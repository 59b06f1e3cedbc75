#!/bin/bash
# A/B over values of one environment variable on one box:
#   bash benchmarks/gpu/ab_env.sh <VAR> "<v1> <v2> ..." <rounds> <script + args>
set -u
var=$1; vals=$2; rounds=$3; shift 3
for r in $(seq 1 "$rounds"); do
  for v in $vals; do
    echo "== round $r: $var=$v"
    env "$var=$v" timeout -k 10 300 python -u "$@" 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
