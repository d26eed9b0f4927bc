#!/bin/bash
# tools/eval_csrmv.sh -- the reference's corpus sweep (eval_csrmv.sh:1-17) for the MI355X drivers: one CSV line per
# Matrix Market file under <mtx dataset dir>, same header, same `--quiet --mtx=` invocation, so existing plotting
# scripts keep working.  Extra arguments go to the driver (e.g. gpu_spmv --device=1 --fp32 --cache).
#   tools/eval_csrmv.sh <mtx dataset dir> <cpu_spmv | gpu_spmv [--device=...]>
if (( $# < 2 )); then
  echo "$0 <mtx dataset dir> <cpu_spmv | gpu_spmv [--device=...]>"
  exit 0
fi
HERE="$(cd "$(dirname "$0")/../merge_spmv_amd" && pwd)"
echo "file, num_rows, num_cols, num_nonzeros, row_length_mean, row_length_std_dev, row_length_variation, row_length_skewness, method_name, setup_ms, avg_spmv_ms, gflops, effective_GBs"
MTX_DIR=$1
shift
DRIVER=$1
shift
find "$MTX_DIR" -name '*.mtx' | sort | while read -r f; do
    "$HERE/$DRIVER" "$@" --quiet --mtx="$f"
done
