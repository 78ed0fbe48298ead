#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
KODHIP_FORCE_BM=256 timeout -k 10 300 python tools/bench_s2_dgrad.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b12_s2_bm256.log
for v in 1 2; do
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-130 
KODHIP_S2_FOLD_MAXC=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-130 
done
