#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L3="s3.b.conv2 128->128 3x3 @40|s2.b.conv2 64->64 3x3 @80|s1.b.conv2 32->32 3x3 @160|s4.b.conv2 256->256 3x3 @20|s2.conv 64->128 3x3s2 @160"
run() { # name env...
  echo "== $1" >> gpurun_out/b2_conv.log; shift
  IFS='|' read -ra LS <<< "$L3"
  env "$@" timeout -k 10 200 python tools/bench_conv.py "${LS[@]}" >> gpurun_out/b2_conv.log 2>&1
}
run nostrip KODHIP_NO_STRIP=1
run strip3 A=1
run strip3_bm128 KODHIP_FORCE_BM=128
run strip5 KODHIP_LIB=tools/ablate/lib_nstb5.so
run strip5_bm128 KODHIP_LIB=tools/ablate/lib_nstb5.so KODHIP_FORCE_BM=128
run strip2 KODHIP_LIB=tools/ablate/lib_nstb2.so
grep -v amdgpu.ids gpurun_out/b2_conv.log | cut -c1-112
timeout -k 10 300 python -m pytest tests/test_hip_postproc.py tests/test_hip_map.py tests/test_hip_frontends.py -m gpu -q -x > gpurun_out/b2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/b2_tests.log; tail -4 gpurun_out/b2_tests.log
timeout -k 10 300 python tools/bench_eval.py > gpurun_out/b2_eval.log 2>&1; tail -2 gpurun_out/b2_eval.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_eval -o eval -- python3 $GRAFT_REPO_ROOT/tools/bench_eval.py > $GRAFT_REPO_ROOT/gpurun_out/b2_prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_eval | head; f=$(ls gpurun_out/prof_eval/*kernel_stats.csv 2>/dev/null | head -1); test -n "$f" && head -15 "$f" | cut -c1-160
