#!/bin/bash
# Quick same-box A/B of the training step (no loop / validation / yv5m legs): tools/ab_quick.sh [--eager] "ENV=..." ...
# ("" = defaults); prints img/s per variant, two alternations.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
extra=""
if [ "$1" = "--eager" ]; then extra="--no-graph"; shift; fi
for rep in 1 2; do
  for v in "" "$@"; do
    r=$(env $v timeout -k 10 300 python bench.py $extra --steps 60 --warmup 10 --no-cpu-baseline --no-loop --no-extra 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || { tail -20 gpurun_out/ab_err.log; exit 1; }
    echo "[$v] $r"
  done
done
