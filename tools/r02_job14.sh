#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time MSPMV_BENCH_FORCE_MG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 2 ) > $O/r2_bench_c5_forced_mg.txt 2>&1
PROFILE_CMD="env PLAN_SKIP_BASE=1 PLAN_BANDS=0 python $GRAFT_REPO_ROOT/tools/plan_bench.py c2" bash tools/gpu_profile.sh r02_plan_c2 > $O/r2_prof_plan2.log 2>&1
