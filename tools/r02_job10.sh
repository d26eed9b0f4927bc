#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
PROFILE_CMD="env SWEEP_DEFAULT_SHAPE=1 python $GRAFT_REPO_ROOT/tools/sweep.py grid3d" bash tools/gpu_profile.sh r02_grid3d > $O/r2_prof_grid3d.log 2>&1
PROFILE_CMD="env SWEEP_DEFAULT_SHAPE=1 python $GRAFT_REPO_ROOT/tools/sweep.py dense5d" bash tools/gpu_profile.sh r02_dense5 > $O/r2_prof_dense5.log 2>&1
