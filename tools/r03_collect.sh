#!/bin/bash
# tools/r03_collect.sh -- the round-3 evidence run on the GPU box (via gpurun): everything lands under gpurun_out/r03/ and is then
# copied into profiles/ (profiles/README.md says which file is which).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r03; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_tests.txt 2>&1
timeout 900 python bench.py > $O/bench_c2.txt 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --dtype f64 --no-cpu-baseline --no-configs > $O/bench_c2_f64.txt 2>&1
timeout 600 python bench.py --steps 50 --warmup 5 --workload dense32 --no-cpu-baseline --no-configs > $O/bench_dense32.txt 2>&1
MSPMV_BENCH_FORCE_MG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench_c5_forced_mg.txt 2>&1
SWEEP_DEFAULT_SHAPE=1 SWEEP_FLAGS=0x40000000 timeout 900 python tools/sweep.py c2 c2d dense32 dense32d dense5d band grid2d grid2d4096 grid3d web rmat c4 2>&1 | grep -v amdgpu.ids > $O/sweep_vs_rocsparse.txt
bash tools/small_driver.sh > $O/small_problem_shapes.txt 2>&1
timeout 600 python tools/first_call.py 2>&1 | grep -v amdgpu.ids > $O/first_call.txt
( python tools/mg_bench.py grid2d 1 2 4 8; python tools/mg_bench.py rmat 1 2 4 8 ) 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/mg_bench.txt
bash tools/run_drivers.sh > $O/drivers.txt 2>&1
timeout 400 python tools/fuzz.py 240 303 2>&1 | tail -3 > $O/fuzz.txt
FUZZ_BIG=0.5 timeout 400 python tools/fuzz.py 120 304 2>&1 | tail -3 >> $O/fuzz.txt
tools/band_ceiling > $O/band_ceiling.txt 2>&1
ls -la $O
