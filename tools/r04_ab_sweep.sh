#!/bin/bash
# same-box A/B of the large sweep only: libmspmv.so vs libmspmv_prev.so, alternating, AB_REPS rounds
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r04; mkdir -p $O
for r in $(seq 1 ${AB_REPS:-3}); do for lib in ${AB_LIBS:-libmspmv.so libmspmv_prev.so}; do
  echo "== $lib"; MSPMV_LIB=merge_spmv_amd/$lib SWEEP_NO_ROCSPARSE=1 SWEEP_DEFAULT_SHAPE=1 timeout 900 python tools/sweep.py ${AB_SWEEP:-dense5d grid3d grid2d4096 dense32d circuit} 2>&1 | grep "DEFAULT" | awk '{printf "%s ", $4} END {print ""}'
done; done > $O/${AB_TAG:-ab_sweep}.txt 2>&1
