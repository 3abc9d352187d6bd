#!/bin/bash
# kernel-time breakdown of the EAGLE-2 draft (scripts/eagle_draft_bench.py) under rocprofv3 --stats; run on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/eagle_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep -o e -- python3 scripts/eagle_draft_bench.py 3 --only-draft > gpurun_out/eagle_prof/log.txt 2>&1
cat gpurun_out/eagle_prof/log.txt | tail -8
python3 scripts/show_stats.py $(find /tmp/ep -name '*kernel_stats.csv' | head -1) | head -40
cp $(find /tmp/ep -name '*kernel_stats.csv' | head -1) gpurun_out/eagle_prof/kernel_stats.csv
python3 scripts/trace_gaps.py $(find /tmp/ep -name '*kernel_trace.csv' | head -1) | tail -22
