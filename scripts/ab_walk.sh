#!/bin/bash
# same-box A/B of two builds of the library on the SAM traversal kernel: scripts/ab/libsamd_hip_{head,new}.so (built by hand, not committed)
#   scripts/ab_walk.sh [rounds]     alternates the two builds, `rounds` times each; prints launch_ms per run
cd "$GRAFT_REPO_ROOT"
L=sam-decoding_amd/samd_hip/libsamd_hip.so
for r in $(seq 1 ${1:-3}); do
  for v in head new; do
    cp scripts/ab/libsamd_hip_$v.so $L
    python3 scripts/walk_probe.py 4194304 1048576 16 30 | tail -1 | python3 -c "import sys; d=eval(sys.stdin.read()); print('$v', d['launch_ms'], round(d['frac'],4))"
  done
done
cp scripts/ab/libsamd_hip_new.so $L
