#!/bin/bash
# round 5: LDS counters of the tile kernels on bench.py's non-lean configurations (tools/run_config.py labels) -- how much of the flags + scan
# path's LDS time is bank conflicts (the lean path's: tools/r05_lds_counters.sh)
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
OUT=$O/lds_counters_general.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for label in ${LABELS:-circuit c3_web c2_f32 c4 dense32}; do
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    d=/tmp/ldsg_${label}_$(echo $set | tr ' ' '_' | cut -c1-30); rm -rf $d
    timeout 400 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "tile_kernel" --output-format csv -d $d -o t -- python $GRAFT_REPO_ROOT/tools/run_config.py $label --steps 10 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -f "$f" ] || { echo "$label [$set]: no counter file" >> $OUT; continue; }
    python3 - "$f" "$label" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    k = k.split("<")[0].split("::")[-1] if "tile_kernel" in k else None
    if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{sys.argv[2]:10s} {k:18s}: " + "  ".join(f"{c} {sum(v)/len(v):.0f}" for c, v in sorted(acc[k].items())))
PY
  done
done
cat $OUT
