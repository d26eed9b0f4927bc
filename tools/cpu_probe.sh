#!/bin/bash
cd "$(dirname "$0")/../merge_spmv_amd" || exit 1
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"; nproc; grep -c processor /proc/cpuinfo
for t in 8 16 32 64 128 256; do echo "## threads $t"; OMP_NUM_THREADS=$t timeout 200 ./cpu_spmv --dense=5 --i=20 --threads=$t 2>&1 | grep -E "Using|avg ms"; done
for t in 32 64 128; do echo "## threads $t, spread"; OMP_PROC_BIND=spread OMP_PLACES=cores timeout 200 ./cpu_spmv --dense=5 --i=20 --threads=$t 2>&1 | grep -E "Using|avg ms"; done
