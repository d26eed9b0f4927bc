#!/bin/bash
# tools/small_driver.sh -- per-call latency of small and mid-size fp64 5-point grids through the C driver (no Python marshalling):
# our stateless call vs rocSPARSE csrmv after analysis.  Output kept as profiles/rNN_small_problem_shapes.txt.
cd "$(dirname "$0")/../merge_spmv_amd" || exit 1
for w in ${SIZES:-30 100 300 500 600 700 800 900 1000 1200 2000}; do
  ./gpu_spmv --grid2d=$w --no-strict --no-hyb --i=2000 2>&1 | awk -v w=$w '
    /num_nonzeros:/ {nz=$2} /^Merge-based CsrMV/ {name="ours (" $4 ")"; sub(/<<<.*/, "", name); name=name ")"} /^rocSPARSE CsrMV/ {name="rocSPARSE csrmv"}
    /^fp64: / { t[name]=$5 } END { printf "grid2d_%-5d nnz %9d:", w, nz; for (k in t) printf "  %s %.1f us", k, t[k]*1000; printf "\n" }'
done
