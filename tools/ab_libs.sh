#!/bin/bash
# tools/ab_libs.sh <libA.so> <libB.so> [workloads...] -- A/B two builds of libmspmv.so on ONE box: boxes differ
# by up to 8 % from each other, so two gpurun calls cannot be compared; this alternates the two libraries
# three times in one call and prints the tile-kernel milliseconds of tools/sweep.py's default shape.
#   make -C merge_spmv_amd && cp merge_spmv_amd/libmspmv.so /tmp/a.so; <edit>; make ...; cp ... tools/_b.so (inside the repo so that it travels)
A=$1; B=$2; shift 2
W=${*:-dense32 dense32d band grid2d rmat c4}
cd "$(dirname "$0")/.." || exit 1
cp merge_spmv_amd/libmspmv.so /tmp/_ab_saved.so
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    cp "$v" merge_spmv_amd/libmspmv.so
    echo -n "$(basename $v): "
    SWEEP_DEFAULT_SHAPE=1 timeout 600 python tools/sweep.py $W 2>&1 | grep DEFAULT | awk '{print $9}' | tr '\n' ' '; echo
  done
done
cp /tmp/_ab_saved.so merge_spmv_amd/libmspmv.so
