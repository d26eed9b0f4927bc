#!/bin/bash
# tools/run_drivers.sh -- the two reference-compatible drivers on the GPU box (C1 on the host cores,
# the reference-flag variants of the GPU configurations); output kept under profiles/.
cd "$(dirname "$0")/../merge_spmv_amd" || exit 1
run() { echo; echo "## $*"; timeout 600 "$@"; }
echo "## host: $(nproc) hardware threads; cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null); cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
lscpu | grep -E 'Model name|Socket|Core|NUMA node\(s\)'
# C1: cpu_spmv --dense=5, fp64; threads default to the cgroup CPU quota; --pin adds the single-socket, first-touch line
run ./cpu_spmv --dense=5 --i=100 --pin
run ./cpu_spmv --grid2d=4096 --i=30 --pin
run ./gpu_spmv --dense=5 --prepared --plan
run ./gpu_spmv --grid2d=4096 --prepared
run ./gpu_spmv --grid3d=200 --prepared
run ./gpu_spmv --dense=32 --size=100000000 --fp32 --prepared
run ./gpu_spmv --wheel=5000000 --fp32
run ./gpu_spmv --quiet --grid3d=200
for w in 30 100 300 700; do run ./gpu_spmv --grid2d=$w --no-strict | grep -E "^## |Merge-based|rocSPARSE|fp64:"; done
