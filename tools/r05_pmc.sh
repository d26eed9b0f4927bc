#!/bin/bash
# round 5: rocprofv3 kernel stats + counter passes for BASELINE config 5 ITSELF -- the stateless call and the
# hot-column plan on the R-MAT scale-26 / 2e9-edge matrix -- with the counters restricted to the kernels of interest
# (--kernel-include-regex), so that the 36 GB generation runs unprofiled.  Also re-collects any other label given.
# usage: bash tools/r05_pmc.sh [label[:mode] ...]        default: c5 c5:hotcols
for spec in ${*:-c5 c5:hotcols}; do
  label=${spec%%:*}; mode=stateless; [[ $spec == *:* ]] && mode=${spec##*:}
  tag=r05_${label}; [ $mode != stateless ] && tag=r05_${label}_${mode}
  steps=20; case $label in c3_web|dense5) steps=100;; rmat24) steps=10;; c5) steps=5;; esac
  dt=f64; case $label in c2_f32|c4|dense32) dt=f32;; esac
  # config 5: rocprofv3 --pmc dies (SIGSEGV in the tool) while torch generates the 2e9 edges, counters restricted or not -- so the matrix is
  # generated ONCE without the profiler, its CSR image parked in RAM-backed /dev/shm (24 GB), and the profiled runs read that
  load=""
  if [ $label = c5 ]; then
    img=/dev/shm/mspmv_c5.img
    [ -f $img ] || python tools/run_config.py c5 --save $img
    load="--load $img"
  fi
  PROFILE_INCLUDE='tile_kernel|hot_' PROFILE_MATCH='mspmv' PROFILE_LABEL=$label$([ $mode != stateless ] && echo _$mode) PROFILE_DTYPE=$dt \
    PROFILE_CMD="python $GRAFT_REPO_ROOT/tools/run_config.py $label --steps $steps --mode $mode $load" \
    timeout 2400 bash tools/gpu_profile.sh $tag > gpurun_out/prof_$tag.log 2>&1
  tail -1 gpurun_out/prof_$tag.log | cut -c1-300
  grep -h "tile_kernel\|hot_" gpurun_out/prof_$tag/*kernel_stats.csv | cut -c1-60,200-330 | head -4
done
rm -f /dev/shm/mspmv_c5.img
