#!/bin/bash
# per-layer conv table (fwd / dgrad / wgrad us, GB/s, TF/s) of one eager step: tools/layer_table.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_lt
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/prof_lt -o lt --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/prof_lt.json 2> gpurun_out/prof_lt.err || exit 1
python3 tools/layer_table.py gpurun_out/prof_lt > gpurun_out/layer_table.txt 2>&1
rm -rf gpurun_out/prof_lt; tail -3 gpurun_out/layer_table.txt
