#!/bin/bash
# round-2 GPU job: full GPU test suite (incl. the full-size C5 test), headline bench, C5 functional run with 8 ranks on
# one device, plan profile + PMC, drivers with the new method lines
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -15 ) > $O/r2_tests_full.txt 2>&1
timeout 600 python bench.py > $O/r2_bench_default.txt 2>&1
PLAN_BANDS=0,4,8 timeout 300 python tools/plan_bench.py c2 c2d > $O/r2_plan_bench2.txt 2>&1
( time MSPMV_BENCH_ONE_DEVICE=1 MSPMV_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 8 --steps 5 --warmup 2 ) > $O/r2_bench_c5_onedev.txt 2>&1
PROFILE_CMD="env PLAN_SKIP_BASE=1 PLAN_BANDS=0 python $GRAFT_REPO_ROOT/tools/plan_bench.py c2" bash tools/gpu_profile.sh r02_plan_c2 > $O/r2_prof_plan.log 2>&1
cd merge_spmv_amd
for args in "--dense=5 --plan --gpus=1,2,8 --mg-one-device" "--grid2d=1000 --fp32 --gpus=4 --mg-one-device --mg-exchange=peer" "--grid2d=300 --gpus=1 --mg-exchange=rccl --no-vendor"; do
  echo "## gpu_spmv $args"; timeout 300 ./gpu_spmv $args --i=200
done > ../$O/r2_drivers_new.txt 2>&1
