#!/bin/bash
# round 5: LDS counters of the tile kernel (and rocSPARSE's) per dispatch -- bank conflicts of the lean reduction's row-strided product reads
# against the LDS instructions' active cycles, on the reference's --dense=<cols> / --grid2d inputs (gpu_spmv driver; separate --pmc passes).
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
OUT=$O/lds_counters.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # label, env assignment or "-", driver arguments...
  label=$1; envkv=$2; shift 2
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    d=/tmp/lds_$(echo "$label $set" | tr ' =' '__' | cut -c1-60); rm -rf $d
    if [ "$envkv" = "-" ]; then timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o t -- $GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv "$@" --no-strict --no-hyb --i=50 > /dev/null 2>&1
    else timeout 300 env "$envkv" rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o t -- $GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv "$@" --no-strict --no-hyb --i=50 > /dev/null 2>&1; fi
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -f "$f" ] || { echo "$label [$set]: no counter file" >> $OUT; continue; }
    python3 - "$f" "$label" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    k = "ours" if "tile_kernel" in k else "rocSPARSE" if "csrmv" in k.lower() else None
    if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{sys.argv[2]:34s} {k:10s}: " + "  ".join(f"{c} {sum(v)/len(v):.0f}" for c, v in sorted(acc[k].items())))
PY
  done
}
run "grid2d_500 fp64 (compact)" - --grid2d=500
run "grid2d_2000 fp64 (general lean)" - --grid2d=2000
run "dense5 fp64 (compact, skinny)" - --dense=5
run "dense8 16M fp64 (compact, skinny)" - --dense=8 --size=16000000
run "dense12 16M fp64 (general lean)" - --dense=12 --size=16000000
run "dense16 16M fp64 (flags)" - --dense=16 --size=16000000
run "dense16 16M fp64 LEAN_AVG=16" MSPMV_LEAN_AVG=16 --dense=16 --size=16000000
cat $OUT
