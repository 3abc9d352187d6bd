#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r05_gputest_head.log; cat gpurun_out/r05_gputest_head.log
SAMD_TEST_POISON=1 timeout 1500 python -m pytest tests/test_gpu_prefill_shaping.py tests/test_gpu_api.py tests/test_gpu_wide_drafts.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r05_gputest_poison_head.log; cat gpurun_out/r05_gputest_poison_head.log
