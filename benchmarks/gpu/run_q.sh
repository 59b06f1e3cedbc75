#!/bin/bash
# session Q: persistent-kernel timing probe, then the 8-rank tensor-parallel engine test
set -o pipefail
mkdir -p gpurun_out
echo "== probe" 
timeout -k 10 240 ./benchmarks/hip/persist_probe 32 10 2>&1 | tee gpurun_out/persist_probe.txt
rc=${PIPESTATUS[0]}
echo "probe rc=$rc"
[ $rc -eq 0 ] || exit $rc
echo "== tp8 engine test"
timeout -k 10 400 python -u -m pytest tests/test_engine_gpu.py -x -v --timeout 300 --timeout-method thread -k "tensor_parallel" 2>&1 | tee gpurun_out/tp8.txt
exit ${PIPESTATUS[0]}
