#!/bin/bash
# round 4: rocprofv3 kernel stats + counter passes for every configuration bench.py reports
# usage: bash tools/r04_pmc.sh [label ...]   labels as in tools/run_config.py; "<label>:plan" / "<label>:hotcols" = the opt-in plans
for spec in ${*:-c2_f32 c2 dense32 dense5 circuit c3_web c3_orkut c4 c2_f32:plan rmat24 rmat24:hotcols}; do
  label=${spec%%:*}; mode=stateless; [[ $spec == *:* ]] && mode=${spec##*:}
  tag=r04_${label}; [ $mode != stateless ] && tag=r04_${label}_${mode}
  steps=20; case $label in c3_web|dense5) steps=100;; rmat24) steps=10;; esac
  dt=f64; case $label in c2_f32|c4|dense32) dt=f32;; esac
  PROFILE_LABEL=$label$([ $mode != stateless ] && echo _$mode) PROFILE_DTYPE=$dt PROFILE_CMD="python $GRAFT_REPO_ROOT/tools/run_config.py $label --steps $steps --mode $mode" \
    timeout 1500 bash tools/gpu_profile.sh $tag > gpurun_out/prof_$tag.log 2>&1
  tail -1 gpurun_out/prof_$tag.log | cut -c1-300
  grep -h "tile_kernel" gpurun_out/prof_$tag/*kernel_stats.csv | cut -c1-60,200-330 | head -2
done
