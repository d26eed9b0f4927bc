#!/bin/bash
# round 4: is a small call's latency instruction fetch?  SQ/SQC instruction-cache counters of the one-launch kernel on small grids
# (separate --pmc passes, kernel trace only), beside rocSPARSE's csrmv kernel in the same driver run.
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(SQC\?_[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_ANY\|WAIT_INST\|INST_LEVEL\|BUSY_CYCLES\|WAVE_CYCLES\|WAVES\|INSTS_SALU\|INSTS_VALU\|INSTS_SMEM\)[A-Z_0-9]*\)" | sort -u | tr '\n' ' ' > $O/icache.txt; echo >> $O/icache.txt
for w in ${SIZES:-100 300}; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM" "SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    d=/tmp/ic_${w}_$(echo $set | tr ' ' '_' | cut -c1-40); rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o t -- $GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv --grid2d=$w --no-strict --no-hyb --i=200 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -f "$f" ] || { echo "grid2d_$w [$set]: no counter file" >> $O/icache.txt; continue; }
    python3 - "$f" "$w" >> $O/icache.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    k = "ours tile_kernel_snap" if "tile_kernel_snap" in k else "rocSPARSE " + k[:40] if "csrmv" in k.lower() else None
    if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"grid2d_{sys.argv[2]} {k}: " + "  ".join(f"{c} {sum(v)/len(v):.0f} (n {len(v)})" for c, v in sorted(acc[k].items())))
PY
  done
done
cat $O/icache.txt
