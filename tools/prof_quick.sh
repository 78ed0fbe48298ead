#!/bin/bash
# usage (on the GPU box): tools/prof_quick.sh <tag>   -> gpurun_out/prof_<tag>/ + family summary on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_$1
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$1 -o $1 --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$1.json 2> gpurun_out/prof_$1.err || exit 1
python3 tools/prof_summary.py gpurun_out/prof_$1 | sed -n '/last full step/,$p' | head -${2:-16}
