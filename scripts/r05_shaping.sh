#!/bin/bash
# the prefill shaping measures: tests, then the summarization line with them
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefill_shaping.py tests/test_gpu_lm_shapes.py tests/test_gpu_wide_drafts.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25
timeout 900 python bench.py --workload summarization --summ-requests 24 > gpurun_out/r05_bench_summarization_shaped.log 2>&1
python3 scripts/show_line.py gpurun_out/r05_bench_summarization_shaped.log summarization 2>/dev/null | head -60 || tail -3 gpurun_out/r05_bench_summarization_shaped.log
