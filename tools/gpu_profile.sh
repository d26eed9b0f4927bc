#!/bin/bash
# tools/gpu_profile.sh <tag> [bench args] -- run on the GPU box (via gpurun):
# rocprofv3 kernel trace + stats of bench.py, then separate PMC passes (never
# combined with sys/hip traces).  Raw output stays in /tmp; only summaries
# (stats CSVs, mspmv kernel rows, per-kernel PMC sums) go to gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_raw_$TAG
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
# PROFILE_MATCH (a regex, default mspmv) selects the kernels whose counters are summarised (e.g. 'mspmv|rocsparse' for the driver);
# PROFILE_CMD overrides the profiled command (e.g. tools/plan_bench.py for the prepared plan); the default is the
# headline bench alone (no cpu_baseline, no `configs`: they launch the same kernel symbols on other matrices)
BENCH=${PROFILE_CMD:-"python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-configs --detail /dev/null $*"}
# PROFILE_INCLUDE (a regex): counters are collected for matching kernels only (--kernel-include-regex) -- what lets a command through
# that first generates tens of GB with thousands of other kernels (BASELINE config 5: tools/r05_pmc.sh)
INC=${PROFILE_INCLUDE:+--kernel-include-regex $PROFILE_INCLUDE}
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o bench -- $BENCH > $OUT/bench_trace.log 2>&1
tail -2 $OUT/bench_trace.log
for f in $(find $RAW/trace -name "*stats*.csv"); do cp $f $OUT/; done
KT=$(find $RAW/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$KT" ]; then head -1 $KT > $OUT/kernel_trace_mspmv.csv; grep -E "${PROFILE_MATCH:-mspmv}" $KT | head -400 >> $OUT/kernel_trace_mspmv.csv; fi
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_READ_SECTORS_sum" "TCP_TCC_READ_REQ_sum"; do
  name=$(echo $pmc | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pmc $INC --output-format csv -d $RAW/pmc_$name -o bench -- $BENCH > $OUT/pmc_$name.log 2>&1
  CC=$(find $RAW/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$CC" ]; then
    python3 - "$CC" "$OUT/pmc_$name.summary.csv" "${PROFILE_MATCH:-mspmv}" <<'PY'
import csv, sys, collections, re
match = re.compile(sys.argv[3])
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r.get("Kernel_Name", "")
    if not match.search(k): continue
    short = k.split("(")[0].split("<")[0].split("::")[-1]
    key = (short, r.get("Counter_Name", ""))
    acc[key][0] += 1; acc[key][1] += float(r.get("Counter_Value", 0) or 0)
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
    for (k, c), (n, s) in sorted(acc.items()):
        f.write(f"{k},{c},{n},{s},{s / n if n else 0}\n")
print(open(sys.argv[2]).read())
PY
  else echo "no counter csv for $pmc"; tail -5 $OUT/pmc_$name.log; fi
done
python3 - "$OUT" "$*" <<'PY'
import csv, json, os, sys
out, extra = sys.argv[1], sys.argv[2]
def avg(name, counter):
    path = os.path.join(out, f"pmc_{name}.summary.csv")
    if not os.path.exists(path): return None
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith("tile_kernel") and r["counter"] == counter: return float(r["avg_per_dispatch"])
    return None
fetch_kb, write_kb = avg("FETCH_SIZE", "FETCH_SIZE"), avg("WRITE_SIZE", "WRITE_SIZE")
dtype = os.environ.get("PROFILE_DTYPE") or ("f64" if "f64" in extra else "f32")
workload = os.environ.get("PROFILE_LABEL") or ("dense32" if "dense32" in extra else "c2" if dtype == "f64" else "c2_f32")      # (bench.py's replay labels)
d = {"workload": workload, "dtype": dtype, "command": os.environ.get("PROFILE_CMD", "bench.py " + extra),
     "collected": "separate rocprofv3 --kernel-trace --pmc passes (tools/gpu_profile.sh), averaged per dispatch of the tile kernel",
     "kernel": "tile kernel of the call (tile_kernel_vec<..,BAND> for column-band candidates, tile_kernel_snap otherwise)",
     "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
     "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; KB -> bytes x1024",
     "tile_kernel_hbm_bytes_per_launch": None if fetch_kb is None or write_kb is None else int((2 * fetch_kb + write_kb) * 1024),
     "TCC_HIT_per_launch": avg("TCC_HIT_sum_TCC_MISS_sum", "TCC_HIT_sum"), "TCC_MISS_per_launch": avg("TCC_HIT_sum_TCC_MISS_sum", "TCC_MISS_sum"),
     "TCC_EA0_RDREQ_128B_per_launch": avg("TCC_EA0_RDREQ_64B_sum_TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_128B_sum")}
json.dump(d, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(d))
PY
ls -la $OUT; du -sh $OUT
