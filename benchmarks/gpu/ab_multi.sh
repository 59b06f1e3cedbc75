#!/bin/bash
# A/B/C of builds of the extension on one box: bash benchmarks/gpu/ab_multi.sh <rounds> "<so1> <so2> ..." <script + args>
# ("default" names the in-tree build).  Prints ms_per_step of each run when the script is bench.py.
set -u
rounds=$1; sos=$2; shift 2
for r in $(seq 1 "$rounds"); do
  for so in $sos; do
    if [ "$so" = default ]; then out=$(timeout -k 10 300 python -u "$@" 2>&1) || exit 1
    else out=$(EVGPT_SO=$PWD/$so timeout -k 10 300 python -u "$@" 2>&1) || exit 1; fi
    echo "round $r $so: $(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
