#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 400 python tools/fuzz.py 120 31 > $O/r2_fuzz3.txt 2>&1
FUZZ_BIG=0.3 timeout 400 python tools/fuzz.py 120 32 >> $O/r2_fuzz3.txt 2>&1
for fm in 2048 8192 16384; do echo "== FUSED_MAX $fm"; MSPMV_FUSED_MAX_TILES=$fm SWEEP_DEFAULT_SHAPE=1 python tools/sweep.py dense5d grid2d web 2>&1 | grep -v "rocSPARSE\|flags 0\|prepared\|amdgpu"; done > $O/r2_fusedmax.txt
