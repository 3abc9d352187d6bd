#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefill_shaping.py tests/test_gpu_llama.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python3 scripts/prefill_probe2.py 512 1237 1333 1536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_prefill_probe3.log
