#!/bin/bash
# r04: launch time of the SAM traversal kernel against the bigram table's sparsity (SAMD_BIGRAM_SLOTS_PER_PAIR), one box
cd "$GRAFT_REPO_ROOT"; ulimit -c 0
for r in 1 2; do
  for p in 2 4 8 16 32; do
    SAMD_BIGRAM_SLOTS_PER_PAIR=$p timeout 120 python3 scripts/walk_probe.py 4194304 1048576 16 30 | tail -1 | python3 -c "import sys; d=eval(sys.stdin.read()); print('slots per pair >= $p:', d['launch_ms'], 'ms', round(d['frac'],4))"
  done
done
