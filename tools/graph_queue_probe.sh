#!/bin/bash
# usage (GPU box): bash tools/graph_queue_probe.sh   -> for several settings: replayed step rate + where the weight gradients run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/gqp; rm -rf $O; mkdir -p $O
i=0
for v in "" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "KODHIP_BRANCH_OVERLAP=0" "KODHIP_WGRAD_FORK=legacy"; do
  i=$((i+1))
  echo "== [$v]"
  for kv in $v; do export "$kv"; done
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-loop --no-extra 2>$O/b$i.err | cut -c1-120 || { tail -5 $O/b$i.err; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$i -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-loop --no-extra > $O/t$i.json 2> $O/t$i.err || { echo "trace failed"; tail -3 $O/t$i.err; }
  python3 tools/wg_schedule.py $O/t$i | tail -3
  for kv in $v; do unset "${kv%%=*}"; done
done
find $O -name "*.db" -delete; find $O -name "*_trace.csv" -size +20M -delete
