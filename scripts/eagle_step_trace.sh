#!/bin/bash
# kernel trace of a short EAGLE-2 bench run: where the GPU idles inside a step (scripts/step_timeline.py); run on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/eagle_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o e -- python3 bench.py --variant eagle2 --model llama3-8b --steps 120 --warmup 20 --no-cpu-baseline > gpurun_out/eagle_trace/log.txt 2>&1
tail -1 gpurun_out/eagle_trace/log.txt | cut -c1-300
python3 scripts/step_timeline.py $(find /tmp/et -name "*kernel_trace.csv" | head -1) 380
