#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/b20_tests.log; cat gpurun_out/b20_tests.log
grep -q " passed" gpurun_out/b20_tests.log || exit 1
if grep -q "failed\|error" gpurun_out/b20_tests.log; then exit 1; fi
for v in 1 2; do
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b20_err.log | cut -c1-130 || { tail -20 gpurun_out/b20_err.log; exit 1; }
done
bash tools/gpu_batch19.sh | head -40
