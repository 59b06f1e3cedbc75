#!/bin/bash
# session Q: persistent-kernel timing probe, then the RCCL capture test
set -o pipefail
mkdir -p gpurun_out
echo "== probe" 
timeout -k 10 240 ./benchmarks/hip/persist_probe 32 10 2>&1 | tee gpurun_out/persist_probe.txt
rc=${PIPESTATUS[0]}
echo "probe rc=$rc"
[ $rc -eq 0 ] || exit $rc
echo "== comm tests"
timeout -k 10 400 python -u -m pytest tests/test_comm_gpu.py -x -v --timeout 240 --timeout-method thread -k "rccl or graph" 2>&1 | tee gpurun_out/comm.txt
exit ${PIPESTATUS[0]}
