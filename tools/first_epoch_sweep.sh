#!/bin/bash
# First-epoch mAP of the HIP trainer under summation-order variants of its kernels, each with bf16 and with fp32
# accumulation of multi-producer activation gradients (KODHIP_DX_FP32).  One process per run (the kernel knobs are read
# once per process).  Output: one "fifths ... {map...}" line per run in $1 (default gpurun_out/fe_sweep.log).
out=${1:-gpurun_out/fe_sweep.log}
mkdir -p "$(dirname "$out")"
variants=("" "KODHIP_FORCE_BM=128" "KODHIP_ROW3=0" "KODHIP_NO_DUAL=1" "KODHIP_S2_FOLD_MAXC=0" "KODHIP_NO_BNRED=1"
          "KODHIP_WGRAD_SLOTS=256" "KODHIP_FORCE_BN=64" "KODHIP_WGRAD_DMA=none" "KODHIP_S2_SEPARATE=1"
          "KODHIP_FORCE_BM=128 KODHIP_ROW3=0" "KODHIP_WGRAD_SLOTS=1024")
for v in "${variants[@]}"; do
  for fp in 0 1; do
    echo "== variant [$v] dx_fp32=$fp" >> "$out"
    env $v KODHIP_DX_FP32=$fp timeout -k 10 300 python3 tools/first_epoch_hip.py >> "$out" 2>&1 || echo "FAILED rc=$?" >> "$out"
  done
done
