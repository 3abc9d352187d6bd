#!/bin/bash
# per-kernel durations of the 16-row forward with and without the L2 warm-up (rocprofv3 --kernel-trace --stats); GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for kb in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp_$kb -o w -- python3 scripts/warm_sweep.py $kb > gpurun_out/r03/warm_prof_$kb.log 2>&1
  cp $(find /tmp/wp_$kb -name '*kernel_stats.csv' | head -1) gpurun_out/r03/warm_kernel_stats_$kb.csv
  echo "== warm $kb KiB"; tail -2 gpurun_out/r03/warm_prof_$kb.log; python3 scripts/show_stats.py gpurun_out/r03/warm_kernel_stats_$kb.csv 10
done
