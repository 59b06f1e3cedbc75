#!/bin/bash
# Counter passes (own runs, kernel-trace only) over the long prefill and the decode step.
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <name> <counters> <program args...>
  local name=$1 ctr=$2; shift 2
  echo "== pmc $name: $ctr"
  timeout -s KILL 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$name -o $name -- python3 "$@" > $ROOT/gpurun_out/pmc_$name.log 2>&1
  local rc=$?
  echo "rc=$rc"; tail -2 $ROOT/gpurun_out/pmc_$name.log
  [ $rc -eq 0 ] || exit $rc
  local f=$(find $ROOT/gpurun_out/pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $ROOT/benchmarks/pmc_summary.py "$f" > $ROOT/gpurun_out/pmc_${name}_summary.txt
  rm -rf $ROOT/gpurun_out/pmc_$name
  head -8 $ROOT/gpurun_out/pmc_${name}_summary.txt | cut -c1-400
}
pass prefill_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $ROOT/benchmarks/bench_prefill.py 4643 &&
pass prefill_fetch "FETCH_SIZE TCC_HIT_sum" $ROOT/benchmarks/bench_prefill.py 4643 &&
pass decode_fetch "FETCH_SIZE TCC_HIT_sum" $ROOT/bench.py --steps 16 --warmup 2 --no-graph &&
pass decode_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU" $ROOT/bench.py --steps 16 --warmup 2 --no-graph
