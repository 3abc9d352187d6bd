#!/bin/bash
# same-box A/B of two builds of the library (scripts/ab/libsamd_hip_{head,new}.so, built by hand, not committed) on any probe command:
#   scripts/ab_lib.sh <rounds> <command ...>       alternates the builds through SAMD_HIP_LIB (the installed library is never touched),
#                                                  prints the command's output per run
# (core dumps off: a faulting build of a 13 GB process filled the box's disk once)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
ulimit -c 0
for v in head new; do
  test -f "scripts/ab/libsamd_hip_$v.so" || { echo "ab_lib.sh: scripts/ab/libsamd_hip_$v.so is missing -- build both variants first" >&2; exit 2; }
done
R=$1; shift
for r in $(seq 1 "$R"); do
  for v in head new; do
    echo "== $v"
    SAMD_HIP_LIB="$PWD/scripts/ab/libsamd_hip_$v.so" timeout 120 "$@" || echo "   (failed: $?)"
  done
done
