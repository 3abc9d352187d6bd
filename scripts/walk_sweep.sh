#!/bin/bash
# k_static_walk_streams: cursors per wave (GPU box).  usage: scripts/walk_sweep.sh <per_wave ...>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
for pw in "$@"; do
  echo "== SAMD_WALK_PER_WAVE=$pw"
  SAMD_WALK_PER_WAVE=$pw timeout 300 python3 scripts/walk_probe.py 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('kernel','launch_ms','lockstep_kernel_ms','frac','achieved','visited_states')})"
done
