#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_network.py tests/test_hip_training.py tests/test_hip_postproc.py -m gpu -q -x --deselect tests/test_hip_training.py::test_first_epoch_map_vs_cpu_trainer > gpurun_out/b4_tests.log 2>&1; echo "rc=$?" >> gpurun_out/b4_tests.log; tail -5 gpurun_out/b4_tests.log | cut -c1-250
timeout -k 10 300 python tools/bench_eval.py > gpurun_out/b4_eval.log 2>&1; tail -1 gpurun_out/b4_eval.log
bash tools/collect_evidence.sh r02a
