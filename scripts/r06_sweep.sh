#!/bin/bash
# r06_sweep.sh [sizes] [dists] [slots] -- the traversal kernel with the round-6 edge blocks and, on the same box, with SAMD_EDGE_BLOCKS=0 (the round-5 edge table)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SIZES=${1:-20,22}; DISTS=${2:-markov,zipf}; SLOTS=${3:-16,4}
timeout 2400 python3 scripts/walk_sweep.py gpurun_out/walk_sweep_blocks.json --sizes $SIZES --dists $DISTS --slots $SLOTS $EXTRA > gpurun_out/r06_walk_sweep_blocks.log 2>&1; echo "blocks rc $?"
if [ -z "$SKIP_TABLE" ]; then SAMD_EDGE_BLOCKS=0 timeout 2400 python3 scripts/walk_sweep.py gpurun_out/walk_sweep_table.json --sizes $SIZES --dists $DISTS --slots $SLOTS $EXTRA > gpurun_out/r06_walk_sweep_table.log 2>&1; echo "table rc $?"; fi
echo "== blocks"; grep "^| 2" gpurun_out/r06_walk_sweep_blocks.log
[ -z "$SKIP_TABLE" ] && { echo "== edge table (r05)"; grep "^| 2" gpurun_out/r06_walk_sweep_table.log; }
