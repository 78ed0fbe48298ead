#!/bin/bash
# 1-rank rehearsal of the N>1 code path (KODHIP_FORCE_COLLECTIVES=1): step rate under the SyncBN transport x bucket placement
# switches, against the plain single-GPU step.  Output: gpurun_out/forced_matrix.log
out=${1:-gpurun_out/forced_matrix.log}; : > "$out"
run() { tag=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --gpus 1 --steps 60 --warmup 8 --no-cpu-baseline --no-loop $EXTRA > /tmp/fm.json 2> /tmp/fm.err || { echo "$tag FAILED" >> "$out"; tail -3 /tmp/fm.err >> "$out"; return; }
  python3 -c "
import json;d=json.loads(open('/tmp/fm.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'], d['engine_options']['syncbn_exchange'], d['engine_options']['comm_overlap'])" >> "$out"; }
run plain X=1
run rccl_inorder KODHIP_FORCE_COLLECTIVES=1 KODHIP_SYNCBN=rccl KODHIP_COMM_OVERLAP=0
run rccl_overlap KODHIP_FORCE_COLLECTIVES=1 KODHIP_SYNCBN=rccl
run peer_inorder KODHIP_FORCE_COLLECTIVES=1 KODHIP_SYNCBN=peer KODHIP_COMM_OVERLAP=0
run peer_overlap KODHIP_FORCE_COLLECTIVES=1 KODHIP_SYNCBN=peer
EXTRA=--no-sync-bn run nosync_overlap KODHIP_FORCE_COLLECTIVES=1
EXTRA=--no-sync-bn run nosync_inorder KODHIP_FORCE_COLLECTIVES=1 KODHIP_COMM_OVERLAP=0
cat "$out"
