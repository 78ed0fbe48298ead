#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_hip_training.py::test_first_epoch_map_vs_cpu_trainer > gpurun_out/b3_tests.log 2>&1; echo "rc=$?" >> gpurun_out/b3_tests.log; tail -12 gpurun_out/b3_tests.log | cut -c1-250
timeout -k 10 300 python tools/bench_variant.py 0.75 0.67 > gpurun_out/b3_yv5m.log 2>&1; tail -1 gpurun_out/b3_yv5m.log
KODHIP_NO_FAST=1 timeout -k 10 300 python tools/bench_variant.py 0.75 0.67 16 > gpurun_out/b3_yv5m_nofast.log 2>&1; tail -1 gpurun_out/b3_yv5m_nofast.log
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b3_bench.log 2>&1; tail -1 gpurun_out/b3_bench.log | cut -c1-300
timeout -k 10 300 python tools/first_epoch_hip.py > gpurun_out/b3_fe.log 2>&1; grep -v amdgpu.ids gpurun_out/b3_fe.log | tail -3
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eval -o eval -- python3 $R/tools/bench_eval.py > $R/gpurun_out/b3_prof.log 2>&1
cd $R; f=$(find gpurun_out/prof_eval -name "*kernel_stats.csv" | head -1); test -n "$f" && head -14 "$f" | cut -c1-170
