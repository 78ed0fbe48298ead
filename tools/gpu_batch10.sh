#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_training.py -m gpu -q -x -k "class_aware" 2>&1 | tail -3
for v in base rs64 rs64n3 nst5; do
  echo "== $v" >> gpurun_out/b10.log
  if [ $v = base ]; then L=""; else L="KODHIP_LIB=tools/ablate/lib_$v.so"; fi
  env $L timeout -k 10 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | cut -c100-160 >> gpurun_out/b10.log
  env $L timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | cut -c40-130 >> gpurun_out/b10.log
done
cat gpurun_out/b10.log
