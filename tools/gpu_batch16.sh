#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in 1 0 1 0; do
KODHIP_BRANCH_OVERLAP=$v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/b16_err_$v.log | cut -c1-130 || { tail -20 gpurun_out/b16_err_$v.log; exit 1; }
done
