#!/bin/bash
# Diagnostic builds of libkodhip.so with compile-time ablation macros (never shipped): tools/build_ablate.sh NAME -DX -DY ...
# -> tools/ablate/lib_NAME.so, load with KODHIP_LIB=tools/ablate/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ablate /tmp/abl_$name
objs=()
for f in object_detection_cib_amd/csrc/*.hip; do
  o=/tmp/abl_$name/$(basename ${f%.hip}).o
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off "$@" -c $f -o $o &
  objs+=($o)
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate/lib_$name.so "${objs[@]}"
echo built tools/ablate/lib_$name.so
