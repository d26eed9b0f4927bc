#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r2_tests_full6.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r2_bench_c2_final2.txt 2>&1
PLAN_BANDS=0,2,4,8 timeout 300 python tools/plan_bench.py c2 > $O/r2_plan_bench3.txt 2>&1
