#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command; only the per-kernel summary comes back (the trace stays in /tmp).
# usage (GPU box): scripts/bench_stats.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r02}; shift
mkdir -p gpurun_out/$tag
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs_$tag -o b -- python3 bench.py "$@" > gpurun_out/$tag/bench_under_rocprof.log 2>&1
cp $(find /tmp/bs_$tag -name '*kernel_stats.csv' | head -1) gpurun_out/$tag/bench_kernel_stats.csv
tail -1 gpurun_out/$tag/bench_under_rocprof.log | cut -c1-160
python3 scripts/show_stats.py gpurun_out/$tag/bench_kernel_stats.csv 14
