#!/bin/bash
# round 4: same-box A/B of merge_spmv_amd/libmspmv.so against merge_spmv_amd/libmspmv_prev.so (built from `git archive HEAD`): small grids by
# graph replay, the large sweep, the block-life trace of the dev build, the instruction-cache counters of a small call, the GPU suite.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r04; mkdir -p $O
F='amdgpu.ids'
T=${AB_TAG:-ab}
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep "passed\|failed\|rror" | tail -5 ) > $O/${T}_gpu_tests.txt 2>&1
{
for r in 1 2; do
  echo "== now"; python tools/small_shapes.py 30 100 300 500 800 1200 2000 2>&1 | grep -v "$F"
  echo "== before"; MSPMV_LIB=merge_spmv_amd/libmspmv_prev.so python tools/small_shapes.py 30 100 300 500 800 1200 2000 2>&1 | grep -v "$F"
done
echo "== C driver (links the new library)"; SIZES="30 100 300 500 1200" bash tools/small_driver.sh
for lib in libmspmv.so libmspmv_prev.so libmspmv.so libmspmv_prev.so; do
  echo "== $lib"; MSPMV_LIB=merge_spmv_amd/$lib SWEEP_NO_ROCSPARSE=1 SWEEP_DEFAULT_SHAPE=1 timeout 900 python tools/sweep.py ${AB_SWEEP:-dense5d grid3d grid2d4096 dense32 dense32d circuit web rmat c4 c2} 2>&1 | grep "^==\|DEFAULT"
done
} > $O/${T}.txt 2>&1
MSPMV_LIB=merge_spmv_amd/libmspmv_dev.so timeout 600 python tools/trace_snap.py g2d100 dense5d 2>&1 | grep -v "$F\|RuntimeWarning\|print(\|res = " > $O/${T}_block_life.txt
SIZES="100" bash tools/r04_icache.sh > /dev/null 2>&1; grep "grid2d_100 ours" $O/icache.txt > $O/${T}_icache.txt
