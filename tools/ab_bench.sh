#!/bin/bash
# A/B of the replayed training step under environment switches, alternating in ONE session on ONE box (boxes differ
# by +-1.5 %): tools/ab_bench.sh "KODHIP_WGRAD_FORK=legacy" "KODHIP_BRANCH_OVERLAP=0" ...   ("" = defaults)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "" "$@"; do
    echo "[$v]"
    env $v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/ab_err.log | cut -c1-110 || { tail -20 gpurun_out/ab_err.log; exit 1; }
  done
done
