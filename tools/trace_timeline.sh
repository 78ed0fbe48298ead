#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_tl
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/prof_tl -o tl --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/prof_tl.json 2> gpurun_out/prof_tl.err || { tail -5 gpurun_out/prof_tl.err; exit 1; }
python3 tools/step_timeline.py gpurun_out/prof_tl 7 --all > gpurun_out/timeline.txt; rm -rf gpurun_out/prof_tl; head -18 gpurun_out/timeline.txt
