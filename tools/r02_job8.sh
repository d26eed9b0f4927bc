#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r2_tests_full5.txt 2>&1
for fl in 0x10000000 0x20000000; do echo "== flags $fl"; SWEEP_FLAGS=$fl SWEEP_DEFAULT_SHAPE=1 python tools/sweep.py c2 dense5d band grid2d grid2d4096 grid3d rmat c4 2>&1 | grep -v "rocSPARSE\|DEFAULT\|prepared\|amdgpu"; done > $O/r2_coords_ab2.txt
cd merge_spmv_amd; for w in 30 100 300 700 1000; do echo "## grid2d $w"; ./gpu_spmv --grid2d=$w --no-strict --i=2000 | grep -E "fp64:" | head -2; done > ../$O/r2_small_driver2.txt 2>&1; cd ..
timeout 300 python tools/fuzz.py 90 22 > $O/r2_fuzz2.txt 2>&1
