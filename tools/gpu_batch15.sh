#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/b15_sweep.log
for bm in 0 128 256; do for bn in 0 32 64 128; do
  KODHIP_FORCE_BM=$bm KODHIP_FORCE_BN=$bn timeout -k 10 120 python tools/sweep_tiles.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/b15_sweep.log || exit 1
done; done
sort -k3,3 -k4,4 -s gpurun_out/b15_sweep.log | head -5
