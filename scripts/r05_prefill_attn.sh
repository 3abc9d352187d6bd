#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o build/pa_probe scripts/probes/prefill_attn_probe.hip || exit 1
{
for wv in ${FORMS:-4 42}; do for a in "200 32 32 0" "512 32 32 0" "1024 32 32 0" "1280 32 32 0" "1333 32 32 0" "1536 32 32 0" "2048 32 32 0" "1536 32 8 0" "700 32 32 300" "1 32 32 2000"; do timeout 200 build/pa_probe $a 20 2048 $wv; done; done
} > gpurun_out/r05_prefill_attn.log 2>&1
cat gpurun_out/r05_prefill_attn.log
