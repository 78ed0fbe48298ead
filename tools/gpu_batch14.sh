#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/b14_tests.log; cat gpurun_out/b14_tests.log
grep -q " passed" gpurun_out/b14_tests.log || exit 1
rm -rf gpurun_out/prof_lt
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/prof_lt -o lt --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/prof_lt.json 2> gpurun_out/prof_lt.err || exit 1
python3 tools/layer_table.py gpurun_out/prof_lt > gpurun_out/layer_table.txt 2>&1
rm -rf gpurun_out/prof_lt
