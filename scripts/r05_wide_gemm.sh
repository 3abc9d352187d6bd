#!/bin/bash
# the hand-written prefill GEMM probe against the library's numbers of profiles/r05_prefill.md
# SHAPES="M N K|M N K|..." MODES="0 1" override the defaults
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value ${DEFS:-} -o build/wide_gemm_probe -Iscripts/probes scripts/probes/wide_gemm_probe.hip || exit 1
SHAPES=${SHAPES:-"256 4096 4096|1024 16384 4096|2048 16384 4096|1536 22016 4096|1536 12288 4096|1536 4096 4096|1536 4096 11008|512 22016 4096|512 12288 4096|1000 22016 4096"}
{
for mode in ${MODES:-0}; do
  IFS='|' read -ra LIST <<< "$SHAPES"
  for shape in "${LIST[@]}"; do
    timeout 120 build/wide_gemm_probe $shape $mode 20 1 ${EPI:-0}
  done
done
} > gpurun_out/r05_wide_gemm.log 2>&1
cat gpurun_out/r05_wide_gemm.log
