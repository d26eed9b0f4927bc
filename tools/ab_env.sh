#!/bin/bash
# tools/ab_env.sh <NAME=VALUE> [workloads...] -- the library with and without one re-tuning variable in the environment, alternated three
# times on ONE box (tile-kernel milliseconds of tools/sweep.py's default shape; see tools/ab_libs.sh for why one box)
KV=$1; shift
W=${*:-dense5d grid3d grid2d4096 grid2d circuit web band dense32 dense32d rmat c4}
cd "$(dirname "$0")/.." || exit 1
echo "# workloads: $W"
for rep in 1 2 3; do
  echo -n "default:  "; SWEEP_DEFAULT_SHAPE=1 timeout 600 python tools/sweep.py $W 2>&1 | grep DEFAULT | awk '{print $9}' | tr '\n' ' '; echo
  echo -n "$KV: "; env "$KV" SWEEP_DEFAULT_SHAPE=1 timeout 600 python tools/sweep.py $W 2>&1 | grep DEFAULT | awk '{print $9}' | tr '\n' ' '; echo
done
