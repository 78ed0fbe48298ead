#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -X faulthandler -m pytest tests/test_hip_ddp.py tests/test_hip_network.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/b23_tests.log; tail -8 gpurun_out/b23_tests.log
grep -q " passed" gpurun_out/b23_tests.log || exit 1
if grep -q "failed\|error" gpurun_out/b23_tests.log; then exit 1; fi
for v in apply legacy apply legacy; do
KODHIP_WGRAD_FORK=$v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b23_err.log | cut -c1-130 || { tail -20 gpurun_out/b23_err.log; exit 1; }
done
