#!/bin/bash
# tools/r04_collect.sh -- the round-4 evidence run on the GPU box (via gpurun): everything lands under gpurun_out/r04/ and is then
# copied into profiles/ (profiles/README.md says which file is which).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r04; mkdir -p $O
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -4 ) > $O/gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" >> $O/gpu_tests.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
timeout 600 python bench.py --steps 50 --warmup 5 --dtype f64 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_c2_f64.json
timeout 600 python bench.py --steps 50 --warmup 5 --workload dense32 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_dense32.json
MSPMV_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --c5-scale 22 --c5-edges 60000000 2>/dev/null | tail -1 > $O/bench_c5_scale22_2ranks_one_device_self_launched.json
SWEEP_DEFAULT_SHAPE=1 SWEEP_FLAGS=0x80000000 timeout 1200 python tools/sweep.py c2 c2d dense32 dense32d dense5d band grid2d grid2d4096 grid3d circuit web rmat orkut c4 2>&1 | grep -v "$F" > $O/sweep_vs_rocsparse.txt
bash tools/small_driver.sh > $O/small_problem_shapes.txt 2>&1
timeout 600 python tools/stream_policy.py 2>&1 | grep -v "$F" > $O/stream_policy.txt
timeout 600 python tools/stream_policy.py c2d dense5d grid2d circuit orkut rmat24 2>&1 | grep -v "$F" >> $O/stream_policy.txt
timeout 600 python tools/first_call.py 2>&1 | grep -v "$F" > $O/first_call.txt
MSPMV_LIB=merge_spmv_amd/libmspmv_dev.so timeout 600 python tools/trace_snap.py dense5d grid3d grid2d4096 dense32d circuit g2d100 2>&1 | grep -v "$F" > $O/block_life.txt
TRACE_FLAGS=0x80000000 MSPMV_LIB=merge_spmv_amd/libmspmv_dev.so timeout 600 python tools/trace_snap.py dense5d grid3d 2>&1 | grep -v "$F" >> $O/block_life.txt
bash tools/run_drivers.sh > $O/drivers.txt 2>&1
( python tools/mg_bench.py grid2d 1 2 4 8; python tools/mg_bench.py rmat 1 2 4 8 ) 2>&1 | grep -v "$F" > $O/mg_bench.txt
PLAN_BANDS=0,8 timeout 600 python tools/plan_bench.py c2 c2d rmat 2>&1 | grep -v "$F" > $O/plan_bench.txt
timeout 400 python tools/fuzz.py 300 401 2>&1 | tail -3 > $O/fuzz.txt
FUZZ_BIG=0.5 timeout 400 python tools/fuzz.py 150 402 2>&1 | tail -3 >> $O/fuzz.txt
ls -la $O
