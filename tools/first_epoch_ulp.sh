#!/bin/bash
# More draws of the HIP trainer's first epoch: the default kernels, element k of the stem's weight one bf16 ulp away (the
# perturbation oracle/first_epoch.py --extra2 gives the CPU trainer's bf16-storage emulation; k = 1 .. $2, default 12).
out=${1:-gpurun_out/fe_ulp.log}
n=${2:-12}
mkdir -p "$(dirname "$out")"; : > "$out"
for k in $(seq 1 $n); do
  echo "== variant [KODHIP_FE_ULP=$k] dx_fp32=0" >> "$out"
  KODHIP_FE_ULP=$k timeout -k 10 300 python3 tools/first_epoch_hip.py >> "$out" 2>&1 || echo "FAILED rc=$?" >> "$out"
done
