#!/bin/bash
# tools/small_driver.sh -- per-call latency of small and mid-size fp64 5-point grids, our stateless call vs rocSPARSE csrmv after analysis.
#   part 1: the C driver in the reference's protocol (gpu_spmv.cu:418-434: one warm-up call, N back-to-back calls between two events),
#           one process per size, the steady level of the loop beside its average (--chunk-times: the loop cut into chunks of 100 calls;
#           boxes stall for ~10 ms now and then, which lands on one method of one size per run);
#   part 2: the same loops INTERLEAVED in one process (tools/ab_driver: ours, ours without the compact front end, rocSPARSE), median of
#           7 loops -- what a 0.1 us difference can be read from.
# Both bound to ONE CPU (CPU=<n>, default 192): below ~300 tiles the loop is bound by the ENQUEUEING THREAD (host enqueue time per call ==
# loop time per call, tools/ab_driver prints both) and which core that thread sits on moves every figure by up to 0.5 us
# (profiles/r05_ab_small_calls.txt).  Output kept as profiles/rNN_small_problem_shapes.txt.
cd "$(dirname "$0")/../merge_spmv_amd" || exit 1
CPU=${CPU:-192}; nproc_all=$(nproc --all); [ "$CPU" -ge "$nproc_all" ] && CPU=0
PIN="taskset -c $CPU"
echo "## part 1: gpu_spmv --grid2d=<w> --i=2000, bound to CPU $CPU; us per call: loop average (steady level = the MEDIAN chunk of 100 calls)"
for w in ${SIZES:-30 100 300 500 600 700 800 900 1000 1200 2000}; do
  $PIN ./gpu_spmv --grid2d=$w --no-strict --no-hyb --i=2000 --chunk-times=100 2>&1 | awk -v w=$w '
    /num_nonzeros:/ {nz=$2} /^Merge-based CsrMV/ {name="ours"} /^rocSPARSE CsrMV/ {name="rocSPARSE"}
    /^fp64: / { t[name]=$5*1000 } /chunk min/ { mn[name]=$9 }
    END { printf "grid2d_%-5d nnz %9d:  ours %.2f (%.2f)  rocSPARSE %.2f (%.2f)  -> %s\n", w, nz, t["ours"], mn["ours"], t["rocSPARSE"], mn["rocSPARSE"], (mn["ours"] <= mn["rocSPARSE"] * 1.01 ? "ours <= rocSPARSE" : "rocSPARSE ahead") }'
done
if [ -x ../tools/ab_driver ]; then
  echo "## part 2: interleaved loops in one process (A = this build, B = the same without the compact front end, R = rocSPARSE), CPU $CPU"
  $PIN ../tools/ab_driver ./libmspmv.so ./libmspmv_dev.so --tune-b=-1 --loops=7 ${SIZES:-30 100 300 500 600 700 800 900 1000 1200 2000}
  echo "## part 2, fp32"
  $PIN ../tools/ab_driver ./libmspmv.so ./libmspmv_dev.so --tune-b=-1 --loops=7 --fp32 ${SIZES:-30 100 300 500 600 700 800 900 1000 1200 2000}
fi
