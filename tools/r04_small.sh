#!/bin/bash
# round 4: small problems -- per-call latency through the C driver and the kernels' own durations (rocprofv3 kernel trace)
bash tools/small_driver.sh
cd /tmp && export TMPDIR=/tmp
for w in 30 100 300 500 600; do
  rm -rf /tmp/ks_$w; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -o t -- $GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv --grid2d=$w --no-strict --no-hyb --i=500 > /dev/null 2>&1
  f=$(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1)
  echo "grid2d_$w kernel durations (ns avg, min):"; grep -i "tile_kernel\|csrmvn" $f | awk -F'","|",|,"' '{n=split($0,a,","); print "   " substr($1,1,60), a[n-5], a[n-4], a[n-3], a[n-2]}' | head -4
done
