#!/bin/bash
# ab_tree.sh <git-rev> <name> -- a self-contained copy of the library and the A/B probes AT <git-rev> under ab_trees/<name>/ (package, bench.py,
# scripts/{attn_ab,step_ab,layer_probe}.py), its libsamd_hip.so built there with the build's own flags.  For same-box A/Bs ACROSS rounds, where
# the C ABI differs and scripts/ab_lib.sh (two .so files under ONE python package) cannot be used: each tree runs its own probes against its
# own library (scripts/ab_trees_run.sh).  ab_trees/ is git-ignored and travels to the GPU box with the snapshot.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
REV=$1; NAME=$2
D=ab_trees/$NAME
rm -rf "$D"; mkdir -p "$D"
git archive "$REV" sam-decoding_amd bench.py scripts/attn_ab.py scripts/step_ab.py scripts/layer_probe.py include | tar x -C "$D"
cd "$D/sam-decoding_amd/csrc"
SRCS=$(ls *.cpp *.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -o ../samd_hip/libsamd_hip.so $SRCS
echo "$REV" > ../../REV
ls -la ../samd_hip/libsamd_hip.so
