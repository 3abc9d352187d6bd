#!/bin/bash
# kernel trace of a short plugin bench run: where the GPU idles inside a step (scripts/step_timeline.py); run on the GPU box.
# usage: scripts/eagle_step_trace.sh [variant] [model] [min kernels per step]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
v=${1:-eagle2}; m=${2:-llama3-8b}; k=${3:-380}
mkdir -p gpurun_out/eagle_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/et_$v -o e -- python3 bench.py --variant $v --model $m --steps 120 --warmup 20 --no-cpu-baseline > gpurun_out/eagle_trace/log_$v.txt 2>&1
python3 scripts/step_timeline.py $(find /tmp/et_$v -name "*kernel_trace.csv" | head -1) $k
