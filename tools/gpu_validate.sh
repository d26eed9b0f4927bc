#!/bin/bash
# tools/gpu_validate.sh [section ...] -- what a round's GPU validation runs on the MI355X box (via gpurun), each section
# writing under gpurun_out/ the files that are then copied into profiles/ (profiles/README.md names the file each
# command produces).  Sections (default: tests bench):
#   tests     the whole -m gpu suite (incl. full-size config 5) + smoke()
#   bench     headline bench lines: c2 (with prepared_plan + cpu_baseline), c2 fp64, dense32, c5 on one GPU,
#             c5 through the N > 1 code path with one rank (RCCL) and with 2 ranks sharing the device (gloo)
#   profiles  rocprofv3 kernel stats + PMC passes: headline c2, dense32, the prepared plan
#   sweeps    every workload vs rocSPARSE, stream-policy and coordinate-pass A/Bs, plan band counts, column-band passes
#   drivers   cpu_spmv / gpu_spmv with the reference's flags, small sizes, the multi-GPU operator on one device
#   ceilings  tools/hw_ceilings.py (gather / stream ceilings, banded probe, scalar-gather probe)
#   ingest    config 3 through the Matrix Market path at com-Orkut size
#   fuzz      randomized differential runs
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out; mkdir -p $O
[ $# -eq 0 ] && set -- tests bench
for section in "$@"; do
case $section in
tests)
  ( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -12 ) > $O/tests.txt 2>&1
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 ;;
bench)
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2.txt 2>&1
  timeout 600 python bench.py --steps 20 --warmup 5 --dtype f64 --no-cpu-baseline > $O/bench_c2_f64.txt 2>&1
  timeout 600 python bench.py --steps 20 --warmup 5 --workload dense32 --no-cpu-baseline > $O/bench_dense32.txt 2>&1
  timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 > $O/bench_c5_n1.txt 2>&1
  MSPMV_BENCH_FORCE_MG=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
      bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench_c5_forced_mg.txt 2>&1
  MSPMV_BENCH_ONE_DEVICE=1 MSPMV_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_c5_2ranks_one_device.txt 2>&1 ;;
profiles)
  bash tools/gpu_profile.sh c2 > $O/prof_c2.log 2>&1
  bash tools/gpu_profile.sh c2_f64 --dtype f64 > $O/prof_c2_f64.log 2>&1
  bash tools/gpu_profile.sh dense32 --workload dense32 > $O/prof_dense32.log 2>&1
  PROFILE_CMD="env PLAN_SKIP_BASE=1 PLAN_BANDS=0 python $PWD/tools/plan_bench.py c2" bash tools/gpu_profile.sh plan_c2 > $O/prof_plan_c2.log 2>&1 ;;
sweeps)
  SWEEP_DEFAULT_SHAPE=1 timeout 900 python tools/sweep.py c2 c2d dense32 dense32d dense5d band grid2d grid2d4096 grid3d web rmat c4 > $O/sweep_vs_rocsparse.txt 2>&1
  for fl in 32 64; do echo "== flags $fl"; SWEEP_FLAGS=$fl SWEEP_DEFAULT_SHAPE=1 python tools/sweep.py c2 rmat band grid2d grid2d4096 grid3d c4 dense32 dense5d 2>&1 | grep -v "rocSPARSE\|DEFAULT\|prepared"; done > $O/stream_policy.txt
  for fl in 0x10000000 0x20000000; do echo "== flags $fl"; SWEEP_FLAGS=$fl SWEEP_DEFAULT_SHAPE=1 python tools/sweep.py c2 dense5d band grid2d grid2d4096 grid3d rmat c4 2>&1 | grep -v "rocSPARSE\|DEFAULT\|prepared"; done > $O/coords_pass.txt
  PLAN_BANDS=0,2,4,8,16 timeout 600 python tools/plan_bench.py c2 c2d rmat > $O/plan_bench.txt 2>&1
  timeout 900 python tools/band_passes_bench.py > $O/band_passes.txt 2>&1 ;;
drivers)
  bash tools/run_drivers.sh > $O/drivers.txt 2>&1
  timeout 300 python tools/small_sizes.py > $O/small_sizes.txt 2>&1
  timeout 600 python tools/mg_bench.py grid2d 1 2 4 8 > $O/mg_bench.txt 2>&1
  timeout 600 python tools/mg_bench.py rmat 1 2 4 8 >> $O/mg_bench.txt 2>&1 ;;
ceilings)
  timeout 900 python tools/hw_ceilings.py policy banded stream scalar > $O/hw_ceilings.txt 2>&1 ;;
ingest)
  ( time timeout 1500 python tools/c3_ingest.py ) > $O/c3_ingest.txt 2>&1 ;;
fuzz)
  timeout 400 python tools/fuzz.py 120 31 > $O/fuzz.txt 2>&1
  FUZZ_BIG=0.3 timeout 400 python tools/fuzz.py 120 32 >> $O/fuzz.txt 2>&1
  FUZZ_BAND=1 FUZZ_BIG=0.1 timeout 400 python tools/fuzz.py 120 33 >> $O/fuzz.txt 2>&1 ;;
*) echo "unknown section $section" ;;
esac
done
