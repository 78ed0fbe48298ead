#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/row3_tests.log; cat gpurun_out/row3_tests.log
grep -q " passed" gpurun_out/row3_tests.log || exit 1
if grep -q "failed\|error" gpurun_out/row3_tests.log; then exit 1; fi
L='s3.b.conv2 128->128 3x3 @40|s2.b.conv2 64->64 3x3 @80|s1.b.conv2 32->32 3x3 @160|s4.b.conv2 256->256 3x3 @20'
for v in 0 2 1; do
echo "== KODHIP_ROW3=$v"
KODHIP_ROW3=$v timeout -k 10 200 python tools/bench_conv.py "s3.b.conv2 128->128 3x3 @40" "s2.b.conv2 64->64 3x3 @80" "s1.b.conv2 32->32 3x3 @160" "s4.b.conv2 256->256 3x3 @20" 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
