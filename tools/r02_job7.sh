#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r2_tests_full4.txt 2>&1
for fl in 0 0x10000000; do echo "== flags $fl"; SWEEP_FLAGS=$fl SWEEP_DEFAULT_SHAPE=1 python tools/sweep.py c2 dense32 dense5d band grid2d grid2d4096 grid3d rmat c4 2>&1 | grep -v "rocSPARSE\|DEFAULT\|prepared\|amdgpu"; done > $O/r2_coords_ab.txt
timeout 300 python tools/fuzz.py 60 21 > $O/r2_fuzz1.txt 2>&1
