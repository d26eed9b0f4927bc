#!/bin/bash
# round 4: the lean short-row reduction (consume_tile_rows) -- full GPU suite, then same-process A/B against the general tile (flags 0x80000000 = MSPMV_TUNE_NO_LEAN)
O=gpurun_out/r04_lean${TAG:-}; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; grep -n "passed\|failed" $O/gpu_tests.txt | tail -3; fi
SWEEP_DEFAULT_SHAPE=${DEFSHAPE:-1} SWEEP_FLAGS=0x80000000 timeout 900 python tools/sweep.py ${W:-dense5d grid2d grid2d4096 grid3d band web rmat c4 dense32d} 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" > $O/sweep.txt
cat $O/sweep.txt
