#!/bin/bash
# kernel breakdown of the shaped prefill at two prompt lengths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in 1333 1536; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prefill2_$n -o p -- python3 scripts/prefill_probe.py $n > gpurun_out/r05_prefill2_$n.log 2>&1
tail -1 gpurun_out/r05_prefill2_$n.log
f=$(find gpurun_out/prefill2_$n -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r05_prefill_shaped_${n}_kernel_stats.csv
python3 scripts/show_stats.py $f 24
done
