#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "stride2 or bn_backward" 2>&1 | tail -3 || exit 1
echo "== longest class first"; timeout -k 10 300 python tools/bench_s2_dgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-60
echo "== interleaved"; KODHIP_S2_INTERLEAVE=1 timeout -k 10 300 python tools/bench_s2_dgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-60
for v in 0 1; do
if [ $v = 1 ]; then export KODHIP_S2_INTERLEAVE=1; fi
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-130
done
