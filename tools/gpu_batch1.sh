#!/bin/bash
# strip kernel validation + timing
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_network.py -m gpu -q -x --deselect tests/test_hip_network.py::test_train_step_well_conditioned_batch > gpurun_out/b1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/b1_tests.log
tail -5 gpurun_out/b1_tests.log
echo "== strip" > gpurun_out/b1_conv.log
timeout -k 10 300 python tools/bench_conv.py >> gpurun_out/b1_conv.log 2>&1
echo "== nostrip" >> gpurun_out/b1_conv.log
KODHIP_NO_STRIP=1 timeout -k 10 300 python tools/bench_conv.py >> gpurun_out/b1_conv.log 2>&1
grep -v amdgpu.ids gpurun_out/b1_conv.log | cut -c1-150
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b1_bench.log 2>&1; tail -1 gpurun_out/b1_bench.log | cut -c1-400
KODHIP_NO_STRIP=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b1_bench_nostrip.log 2>&1; tail -1 gpurun_out/b1_bench_nostrip.log | cut -c1-200
for v in "" "KODHIP_FORCE_BM=128" "KODHIP_NO_BNRED=1" "KODHIP_NO_STRIP=1"; do
  echo "== first epoch [$v]" >> gpurun_out/b1_fe.log
  env $v timeout -k 10 200 python tools/first_epoch_hip.py >> gpurun_out/b1_fe.log 2>&1
done
grep -v amdgpu.ids gpurun_out/b1_fe.log
