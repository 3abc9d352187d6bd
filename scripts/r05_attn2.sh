#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
bash scripts/r05_prefill_attn.sh | grep "rows\|check" | paste - - | awk '{print $2,$3,$4,$14,$15,$16,$17,$18,$19,$20,$21,$22,$23,$24}'
timeout 900 python -m pytest tests/test_gpu_prefill_shaping.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
timeout 600 python3 scripts/prefill_probe2.py 512 1237 1333 1536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_prefill_probe4.log
SAMD_PREFILL_ATTENTION=sdpa timeout 600 python3 scripts/prefill_probe2.py 512 1333 1536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_prefill_probe4_sdpa.log
