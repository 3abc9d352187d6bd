#!/bin/bash
# where does the prompt-attention kernel's time go?  Diagnostic builds with parts switched off (results wrong, timing only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
{
for abl in 0 1 2 4 6 8 9 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -DPA_ABL=$abl -o build/pa_abl scripts/probes/prefill_attn_probe.hip || exit 1
  for a in "1536 32 32 0" "1 32 32 2000"; do echo -n "PA_ABL=$abl  "; timeout 100 build/pa_abl $a 20 2048 42 | grep "^rows"; done
done
} > gpurun_out/r05_prefill_attn_ablate.log 2>&1
cat gpurun_out/r05_prefill_attn_ablate.log
