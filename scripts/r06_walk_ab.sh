#!/bin/bash
# r06_walk_ab.sh [sizes] [dists] [slots] -- same-box A/B of two builds of the library (scripts/ab/libsamd_hip_{head,new}.so) on the traversal-kernel sweep (no counters)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SIZES=${1:-20,22}; DISTS=${2:-zipf,markov}; SLOTS=${3:-16,4}
for r in 1 2; do for v in head new; do
  echo "== $v"
  SAMD_HIP_LIB="$PWD/scripts/ab/libsamd_hip_$v.so" timeout 1200 python3 scripts/walk_sweep.py gpurun_out/walk_ab_$v.json --sizes $SIZES --dists $DISTS --slots $SLOTS --no-pmc 2>&1 | grep "^| 2"
done; done
