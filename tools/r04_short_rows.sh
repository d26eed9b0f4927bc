#!/bin/bash
# round 4: the lean short-row tile beside rocSPARSE's analysed kernel in one process (gpu_spmv driver): durations + SQ counters
for spec in "dense5 --dense=5" "grid3d --grid3d=200"; do
  set -- $spec; name=$1; flag=$2
  PROFILE_MATCH='mspmv|rocsparse' PROFILE_CMD="$GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv $flag --i=300" bash tools/gpu_profile.sh r04sr_$name > gpurun_out/prof_r04sr_$name.log 2>&1
  PROFILE_MATCH='tile_kernel|csrmvn' PROFILE_CMD="$GRAFT_REPO_ROOT/merge_spmv_amd/gpu_spmv $flag --i=300" bash tools/gpu_pmc_sq.sh r04sr_$name > gpurun_out/sq_r04sr_$name.txt 2>&1
  cat gpurun_out/sq_r04sr_$name.txt
  grep -i "tile_kernel\|csrmvn" gpurun_out/prof_r04sr_$name/*stats*.csv | head -6
done
