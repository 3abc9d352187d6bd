#!/bin/bash
# kernel trace of a short default bench run -> GPU idle gaps (scripts/step_timeline.py); run on the GPU box.  usage: bench_trace.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bench_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o b -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/bench_trace/log.txt 2>&1
tail -1 gpurun_out/bench_trace/log.txt | cut -c1-200
python3 scripts/step_timeline.py $(find /tmp/bt -name "*kernel_trace.csv" | head -1) 250
python3 scripts/request_timeline.py $(find /tmp/bt -name "*kernel_trace.csv" | head -1)
