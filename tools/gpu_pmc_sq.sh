#!/bin/bash
# tools/gpu_pmc_sq.sh <tag> [bench args] -- SQ-side counters of the tile kernel (separate passes).
set -u
TAG=${1:-sq}; shift || true
BARGS="$*"
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
RAW=/tmp/sq_raw_$TAG
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
# PROFILE_CMD overrides the profiled command (e.g. the gpu_spmv driver); PROFILE_MATCH (regex, default tile_kernel) selects the kernels
BENCH=${PROFILE_CMD:-"python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs $BARGS"}
i=0
PMCS=${PMCS:-full}
if [ "$PMCS" = "lds" ]; then
  set -- "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
else
  set -- "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
fi
for pmc in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $RAW/p$i -o bench -- $BENCH > $OUT/p$i.log 2>&1
  CC=$(find $RAW/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$CC" ]; then
    python3 - "$CC" "${PROFILE_MATCH:-tile_kernel}" <<'PY'
import csv, sys, collections, re
match = re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if not match.search(k): continue
    short = k.split("(")[0].split("<")[0].split("::")[-1]
    key = (short, r.get("Counter_Name", ""))
    acc[key][0] += 1; acc[key][1] += float(r.get("Counter_Value", 0) or 0)
for (k, c), (n, s) in sorted(acc.items()):
    print(f"{k:26s} {c:28s} avg/dispatch {s / n if n else 0:16.1f}   (n={n})")
PY
  else echo "no csv for: $pmc"; tail -3 $OUT/p$i.log; fi
done
