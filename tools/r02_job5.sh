#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r2_tests_full2.txt 2>&1
bash tools/gpu_profile.sh r02_c2 > $O/r2_prof_c2.log 2>&1
bash tools/gpu_profile.sh r02_dense32 --workload dense32 > $O/r2_prof_d32.log 2>&1
SWEEP_DEFAULT_SHAPE=1 timeout 900 python tools/sweep.py c2 c2d dense32 dense32d dense5d band grid2d grid2d4096 grid3d web rmat c4 > $O/r2_sweep_vs_rocsparse.txt 2>&1
timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 > $O/r2_bench_c5_n1.txt 2>&1
( time MSPMV_BENCH_ONE_DEVICE=1 MSPMV_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 ) > $O/r2_bench_c5_onedev2.txt 2>&1
timeout 600 python tools/hw_ceilings.py policy banded stream > $O/r2_hw_ceilings.txt 2>&1
bash tools/run_drivers.sh > $O/r2_drivers.txt 2>&1
timeout 300 python tools/small_sizes.py > $O/r2_small_sizes.txt 2>&1
timeout 600 python tools/mg_bench.py grid2d 1 2 4 8 > $O/r2_mg_bench.txt 2>&1
timeout 600 python tools/mg_bench.py rmat 1 2 4 8 >> $O/r2_mg_bench.txt 2>&1
( time timeout 1500 python tools/c3_ingest.py ) > $O/r2_c3_ingest.txt 2>&1
