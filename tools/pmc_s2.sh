#!/bin/bash
# usage (GPU box): bash tools/pmc_s2.sh   -> gpurun_out/pmc_s2/<set>/...counter_collection.csv (+ kernel traces), one rocprofv3 run per counter set
R="$GRAFT_REPO_ROOT"; test -n "$R" || R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/pmc_s2"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  test -n "$set" || continue
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/set$i" -o p -- python3 "$R/tools/pmc_s2.py" > "$O/set$i.out" 2> "$O/set$i.err" \
    && echo "set$i ok: $set" || { echo "set$i FAILED: $set"; tail -3 "$O/set$i.err"; }
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum
SETS
find "$O" -name "*.db" -delete
du -sh "$O"
