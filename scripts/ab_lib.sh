#!/bin/bash
# same-box A/B of two builds of the library (scripts/ab/libsamd_hip_{head,new}.so, built by hand, not committed) on any probe command:
#   scripts/ab_lib.sh <rounds> <command ...>       alternates the builds, prints the command's output per run; leaves `new` installed
# (core dumps off: a faulting build of a 13 GB process filled the box's disk once)
cd "$GRAFT_REPO_ROOT"
ulimit -c 0
L=sam-decoding_amd/samd_hip/libsamd_hip.so
R=$1; shift
for r in $(seq 1 $R); do
  for v in head new; do
    cp scripts/ab/libsamd_hip_$v.so $L
    echo "== $v"; timeout 120 "$@" || echo "   (failed: $?)"
  done
done
cp scripts/ab/libsamd_hip_new.so $L
