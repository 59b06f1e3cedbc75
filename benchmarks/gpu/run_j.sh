#!/bin/bash
# GPU session J: decode step time of GEMV build variants (EVGPT_SO selects the library), two rounds.
set -u
mkdir -p gpurun_out
ROOT=$PWD
for round in 1 2; do
  for v in default wf occ1 wfocc1 r1 simple; do
    so=$ROOT/eventgpt_amd/_C_$v.so
    [ $v = default ] && so=$ROOT/eventgpt_amd/_C.so
    [ -f $so ] || continue
    EVGPT_SO=$so timeout -k 10 200 python bench.py --steps 192 --warmup 16 > gpurun_out/bj_${v}_$round.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "variant $v rc=$rc"; tail -5 gpurun_out/bj_${v}_$round.log; exit $rc; fi
    echo "$v round $round: $(python -c "import json,sys; d=json.loads(open('gpurun_out/bj_${v}_$round.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['event_prefill_ms'])")"
  done
done
