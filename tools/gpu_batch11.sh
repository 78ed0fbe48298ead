#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/b11.log
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "dual" 2>&1 | tail -15 &&
timeout -k 10 600 python -m pytest tests/test_hip_network.py tests/test_hip_ddp.py -m gpu -q -x 2>&1 | tail -5 &&
for v in dual nodual dual nodual; do
  echo "== $v" >> gpurun_out/b11.log
  if [ $v = dual ]; then L="KODHIP_X=0"; else L="KODHIP_NO_DUAL=1"; fi
  env $L timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-130 >> gpurun_out/b11.log
done
cat gpurun_out/b11.log
