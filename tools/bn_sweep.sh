#!/bin/bash
# Launch-shape sweep of the two apply passes (tools/bench_bn.py under the library's A/B knobs) -> profiles/r06_bn_apply_sweep.txt.
# tools/ablate/lib_base.so = tools/build_ablate.sh base at the commit to compare against.
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/bench_bn.py 2>&1 | grep -E "^\[|^sum"; }
[ -f tools/ablate/lib_base.so ] && run KODHIP_LIB=tools/ablate/lib_base.so
run X=default
for u in 2 4; do for g in 12800 32768 1000000; do run KODHIP_BN_LDS=-1 KODHIP_BN_U=$u KODHIP_BN_BLOCK=256 KODHIP_BN_GRID=$g; done; done
for b in 256 512 1024; do for g in 16384 1000000; do run KODHIP_BN_LDS=1 KODHIP_BN_U=1 KODHIP_BN_BLOCK=$b KODHIP_BN_GRID=$g; done; done
run KODHIP_BN_LDS=1 KODHIP_BN_U=2 KODHIP_BN_BLOCK=256 KODHIP_BN_GRID=1000000
run BENCH_BN_SHAPES=yv5m
run BENCH_BN_SHAPES=yv5m KODHIP_BN_LDS=-1 KODHIP_BN_U=2 KODHIP_BN_GRID=32768
