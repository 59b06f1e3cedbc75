#!/bin/bash
set -o pipefail
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
echo "== harness chain"
timeout -k 10 900 bash benchmarks/harness_chain.sh /tmp/evgpt_chain 8 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/harness_chain.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
