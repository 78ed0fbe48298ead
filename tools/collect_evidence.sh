#!/bin/bash
# usage (on the GPU box): bash tools/collect_evidence.sh <tag> [variant]      (variant: yv5s (default) | yv5m = BASELINE configs[4])
# Collects, under gpurun_out/ev_<tag>/: the bench line, a kernel trace with stats, the FETCH_SIZE / WRITE_SIZE PMC
# passes (separate runs, kernel trace only - MI355X_MICROARCH.md, HBM) and an MFMA counter pass.  rocprofv3 gets the
# interpreter directly after `--` (no wrapper).  tools/refresh_profiles.py turns the directory into profiles/<tag>_*.
tag=${1:-r06}
variant=${2:-yv5s}
V="--variant $variant"
R="$GRAFT_REPO_ROOT"; test -n "$R" || R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/ev_$tag"; rm -rf "$O"; mkdir -p "$O"
cd "$R" && timeout -k 10 400 python3 bench.py $V > "$O/bench.json" 2> "$O/bench.err" || { echo "bench failed"; tail -5 "$O/bench.err"; exit 1; }
tail -1 "$O/bench.json" | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python3 "$R/bench.py" $V --steps 5 --warmup 3 --no-cpu-baseline --no-loop > "$O/trace.json" 2> "$O/trace.err" || { echo "trace failed"; tail -5 "$O/trace.err"; exit 1; }
echo "trace ok"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -o p -- python3 "$R/bench.py" $V --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-loop --no-extra > "$O/pmc_$c.json" 2> "$O/pmc_$c.err" || { echo "pmc $c failed"; tail -5 "$O/pmc_$c.err"; exit 1; }
  echo "pmc $c ok"
done
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -o p -- python3 "$R/bench.py" $V --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-loop --no-extra > "$O/pmc_mfma.json" 2> "$O/pmc_mfma.err" || { echo "pmc mfma failed"; tail -5 "$O/pmc_mfma.err"; }
echo "pmc mfma done"
# keep the merge small: counter csvs can be large
find "$O" -name "*.db" -delete
du -sh "$O"
