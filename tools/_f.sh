cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
SWEEP_DEFAULT_SHAPE=1 SWEEP_FLAGS=0x40000000 timeout 900 python tools/sweep.py c2 c2d dense32 dense32d dense5d band grid2d grid2d4096 grid3d web rmat c4 2>&1 | grep -v amdgpu.ids > $O/sweep_vs_rocsparse.txt
bash tools/small_driver.sh > $O/small_problem_shapes.txt 2>&1
timeout 600 python tools/first_call.py 2>&1 | grep -v amdgpu.ids > $O/first_call.txt
timeout 900 python bench.py > $O/bench_c2.txt 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --workload dense32 --no-cpu-baseline --no-configs > $O/bench_dense32.txt 2>&1
bash tools/run_drivers.sh > $O/drivers.txt 2>&1
