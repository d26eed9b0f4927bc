#!/bin/bash
# round 5: SQ counters of a small call -- the compact front end, the general kernel (MSPMV_COMPACT_MAX_TILES=0) and rocSPARSE's csrmv kernel
# in the same driver runs, per dispatch: instructions by kind, busy / wait cycles, instruction-cache misses (separate --pmc passes,
# kernel trace only).  Sizes: 28, 251 and 697 tiles (one block per CU at most ... 2.7 per CU).
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
OUT=$O/small_call_counters.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for w in ${SIZES:-100 300 500}; do
  for variant in compact general; do
    [ $variant = general ] && export MSPMV_COMPACT_MAX_TILES=0 || unset MSPMV_COMPACT_MAX_TILES
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
      d=/tmp/ic_${w}_${variant}_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
      timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o t -- $GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv --grid2d=$w --no-strict --no-hyb --i=200 > /dev/null 2>&1
      f=$(find $d -name "*counter_collection.csv" | head -1)
      [ -f "$f" ] || { echo "grid2d_$w $variant [$set]: no counter file" >> $OUT; continue; }
      python3 - "$f" "$w" "$variant" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    k = f"ours ({sys.argv[3]})" if "tile_kernel_snap" in k else "rocSPARSE csrmv" if ("csrmv" in k.lower() and sys.argv[3] == "compact") else None
    if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"grid2d_{sys.argv[2]} {k:16s}: " + "  ".join(f"{c} {sum(v)/len(v):.0f}" for c, v in sorted(acc[k].items())))
PY
    done
  done
done
unset MSPMV_COMPACT_MAX_TILES
cat $OUT
