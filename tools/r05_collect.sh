#!/bin/bash
# tools/r05_collect.sh -- the round-5 evidence run on the GPU box (via gpurun): everything lands under gpurun_out/r05/ and is then
# copied into profiles/ (profiles/README.md says which file is which).  Every step under its own timeout.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05; mkdir -p $O
F='amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl'
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -v "$F" | tail -4 ) > $O/gpu_tests.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" >> $O/gpu_tests.txt
timeout 1200 python bench.py 2>$O/bench_c2.err | tail -1 > $O/bench_c2.json
timeout 600 python bench.py --steps 50 --warmup 5 --dtype f64 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_c2_f64.json
timeout 600 python bench.py --steps 50 --warmup 5 --workload dense32 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_dense32.json
MSPMV_BENCH_FORCE_MG=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 1 --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_c5_forced_mg.json
MSPMV_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --c5-scale 22 --c5-edges 60000000 2>/dev/null | tail -1 > $O/bench_c5_scale22_2ranks_one_device_self_launched.json
timeout 900 bash tools/small_driver.sh > $O/small_problem_shapes.txt 2>&1
SWEEP_DEFAULT_SHAPE=1 SWEEP_FLAGS=0x80000000 timeout 1200 python tools/sweep.py c2 c2d dense32 dense32d dense5d band grid2d grid2d4096 grid3d circuit web rmat orkut c4 2>&1 | grep -v "$F" > $O/sweep_vs_rocsparse.txt
timeout 600 bash tools/run_drivers.sh > $O/drivers.txt 2>&1
PLAN_BANDS=0,8 timeout 600 python tools/plan_bench.py c2 c2d 2>&1 | grep -v "$F" > $O/plan_bench.txt
timeout 400 python tools/fuzz.py 300 501 2>&1 | tail -3 > $O/fuzz.txt
FUZZ_BIG=0.5 timeout 400 python tools/fuzz.py 150 502 2>&1 | tail -3 >> $O/fuzz.txt
# the headline under rocprofv3 --kernel-trace --stats (+ its counter passes): what roofline.achieved is cross-checked against
PROFILE_LABEL=c2_f32 PROFILE_DTYPE=f32 timeout 900 bash tools/gpu_profile.sh r05_c2_f32 > $O/prof_c2_f32.log 2>&1
ls -la $O
