#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_network.py tests/test_hip_ddp.py tests/test_hip_training.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/b21_tests.log; cat gpurun_out/b21_tests.log
grep -q " passed" gpurun_out/b21_tests.log || exit 1
if grep -q "failed\|error" gpurun_out/b21_tests.log; then exit 1; fi
for v in dgrad legacy dgrad legacy; do
KODHIP_WGRAD_FORK=$v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b21_err.log | cut -c1-130 || { tail -20 gpurun_out/b21_err.log; exit 1; }
done
KODHIP_WGRAD_OVERLAP=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b21_err.log | cut -c1-130
bash tools/gpu_batch19.sh | head -18
