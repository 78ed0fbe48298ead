#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
for v in 1 2; do
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b22_err.log | cut -c1-130 || { tail -20 gpurun_out/b22_err.log; exit 1; }
done
KODHIP_BRANCH_OVERLAP=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b22_err.log | cut -c1-130
bash tools/gpu_batch19.sh | head -16
