#!/bin/bash
# tools/run_drivers.sh -- the two reference-compatible drivers on the GPU box (C1 on the host cores,
# the reference-flag variants of the GPU configurations); output kept under profiles/.
cd "$(dirname "$0")/../merge_spmv_amd" || exit 1
run() { echo; echo "## $*"; timeout 300 "$@"; }
echo "## host: $(nproc) hardware threads"; lscpu | grep -E 'Model name|Socket|Core|NUMA node\(s\)'
run ./cpu_spmv --dense=5 --i=50
run ./cpu_spmv --grid2d=4096 --i=30
run ./gpu_spmv --dense=5
run ./gpu_spmv --grid2d=4096
run ./gpu_spmv --dense=32 --size=100000000 --fp32
run ./gpu_spmv --wheel=5000000 --fp32
run ./gpu_spmv --quiet --grid3d=200
