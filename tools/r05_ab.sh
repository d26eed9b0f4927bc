#!/bin/bash
# tools/r05_ab.sh -- small-call latency on the GPU box: this build (A) vs another (B: PREV=<lib>, default the previous build) or vs
# its own general kernel (compact front end switched off) vs rocSPARSE (R), interleaved loops in one process (tools/ab_driver), the
# process bound to one CPU (the loop is bound by the enqueueing thread for small grids, and which core that is moves it by 0.5 us).
cd "$(dirname "$0")/.." || exit 1
CPUS=${CPUS:-"192 0"}
for c in $CPUS; do
  echo "## taskset -c $c: A = this build, B = ${PREV:-merge_spmv_amd/libmspmv_prev.so}"
  taskset -c $c tools/ab_driver merge_spmv_amd/libmspmv.so ${PREV:-merge_spmv_amd/libmspmv_prev.so} --loops=7 ${SIZES:-30 100 300 500 600 700}
  echo "## taskset -c $c: A = compact front end, B = general kernel (same build)"
  taskset -c $c tools/ab_driver merge_spmv_amd/libmspmv.so merge_spmv_amd/libmspmv_dev.so --tune-b=-1 --loops=7 ${SIZES:-30 100 300 500 600 700}
  echo "## taskset -c $c: fp32, A = compact front end, B = general kernel (same build)"
  taskset -c $c tools/ab_driver merge_spmv_amd/libmspmv.so merge_spmv_amd/libmspmv_dev.so --tune-b=-1 --loops=7 --fp32 ${SIZES:-30 100 300 500 600 700}
done
