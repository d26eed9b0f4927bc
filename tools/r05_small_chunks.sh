#!/bin/bash
# tools/r05_small_chunks.sh -- the reference-protocol loop of gpu_spmv (one warm-up call, N back-to-back calls between two events)
# on the 11 small/mid grid sizes, RUNS times over, each method followed by the same loop cut into chunks of 100 calls
# (--chunk-times): where a loop average that moves from run to run comes from.  Output: profiles/r05_small_chunk_times.txt
cd "$(dirname "$0")/../merge_spmv_amd" || exit 1
for run in $(seq 1 ${RUNS:-3}); do
  echo "== run $run"
  for w in ${SIZES:-30 100 300 500 600 700 800 900 1000 1200 2000}; do
    ./gpu_spmv --grid2d=$w --no-strict --no-hyb --i=2000 --chunk-times=100 2>&1 | awk -v w=$w '
      /num_nonzeros:/ {nz=$2}
      /^Merge-based CsrMV/ {name="ours"} /^rocSPARSE CsrMV/ {name="rocsparse"}
      /^fp64: / { t[name]=$5*1000 }
      /chunk times/ { line[name]=$0; sub(/.*chunk\):/, "", line[name]) }
      /chunk min/ { mn[name]=$3; mx[name]=$6 }
      END { printf "grid2d_%-5d nnz %9d: ours %.2f us (chunks %.2f..%.2f)  rocsparse %.2f us (chunks %.2f..%.2f)\n", w, nz, t["ours"], mn["ours"], mx["ours"], t["rocsparse"], mn["rocsparse"], mx["rocsparse"];
            if (mx["ours"] > 1.3*mn["ours"]) printf "    ours chunks:%s\n", line["ours"];
            if (mx["rocsparse"] > 1.3*mn["rocsparse"]) printf "    rocsparse chunks:%s\n", line["rocsparse"]; }'
  done
done
