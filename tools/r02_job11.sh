#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
for fl in 0 0x7000000 0x8000000 0x9000000 0xA000000 0xB000000; do echo "== flags $fl"; SWEEP_FLAGS=$fl SWEEP_DEFAULT_SHAPE=1 python tools/sweep.py grid3d grid2d4096 band c4 rmat c2 dense32 dense5d 2>&1 | grep -v "rocSPARSE\|DEFAULT\|prepared\|amdgpu"; done > $O/r2_runlen.txt
