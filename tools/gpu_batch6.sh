#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_training.py::test_first_epoch_map_vs_cpu_trainer > gpurun_out/b6_tests.log 2>&1; echo "rc=$?" >> gpurun_out/b6_tests.log; tail -6 gpurun_out/b6_tests.log | cut -c1-250
timeout -k 10 300 python tools/bench_conv.py > gpurun_out/b6_conv.log 2>&1; grep -v amdgpu.ids gpurun_out/b6_conv.log | cut -c1-160
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200
timeout -k 10 300 python tools/bench_eval.py 2>&1 | tail -1
timeout -k 10 300 python tools/bench_eval.py --eager 2>&1 | tail -1
timeout -k 10 300 python tools/bench_variant.py 0.75 0.67 2>&1 | tail -1
