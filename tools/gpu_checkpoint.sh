#!/bin/bash
# full GPU test suite, then the evidence collection (tools/collect_evidence.sh <tag>); stops at the first failure
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/ckpt_tests.log; cat gpurun_out/ckpt_tests.log
grep -q " passed" gpurun_out/ckpt_tests.log || exit 1
if grep -q "failed\|error" gpurun_out/ckpt_tests.log; then exit 1; fi
bash tools/collect_evidence.sh ${1:-r03}
