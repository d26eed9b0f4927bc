#!/bin/bash
# round 4: kernel-argument preload + one-round-trip prologue of the one-launch kernel -- same-box A/B against the library of the commit before
# (merge_spmv_amd/libmspmv_prev.so, built from `git archive HEAD~`), small grids by graph replay, the large sweep, the block-life trace.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r04; mkdir -p $O
F='amdgpu.ids'
{
for r in 1 2; do
  echo "== now"; python tools/small_shapes.py 30 100 300 500 800 1200 2000 2>&1 | grep -v "$F"
  echo "== before"; MSPMV_LIB=merge_spmv_amd/libmspmv_prev.so python tools/small_shapes.py 30 100 300 500 800 1200 2000 2>&1 | grep -v "$F"
done
echo "== C driver (links the new library)"; SIZES="30 100 300 500 1200" bash tools/small_driver.sh
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 SIZES="30 100 300" bash tools/small_driver.sh
echo "== HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 SIZES="30 100 300" bash tools/small_driver.sh
} > $O/preload_ab.txt 2>&1
for lib in libmspmv.so libmspmv_prev.so; do
  echo "== $lib"; MSPMV_LIB=merge_spmv_amd/$lib SWEEP_NO_ROCSPARSE=1 SWEEP_DEFAULT_SHAPE=1 timeout 900 python tools/sweep.py dense5d grid3d grid2d4096 dense32 dense32d circuit web c4 2>&1 | grep "^==\|DEFAULT"
done >> $O/preload_ab.txt 2>&1
MSPMV_LIB=merge_spmv_amd/libmspmv_dev.so timeout 600 python tools/trace_snap.py g2d100 dense5d web rmat 2>&1 | grep -v "$F" > $O/block_life_preload.txt
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 ) > $O/gpu_tests_preload.txt 2>&1
