#!/bin/bash
# round-4 closing measurements at HEAD (GPU box): default bench x2, the driver's window, rocprofv3 kernel stats of the bench, the 8-row layer's
# kernel stats, the variant lines.  Every step under its own timeout; core dumps off.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
timeout 400 python3 bench.py > $O/bench_default.log 2>&1
timeout 400 python3 bench.py --no-cpu-baseline > $O/bench_default_rep1.log 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_20steps.log 2>&1
timeout 500 scripts/bench_stats.sh final_stats --no-cpu-baseline --no-live-pmc > $O/bench_stats.txt 2>&1
mkdir -p $O/layer8
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/layer8/trace -o t -- python3 scripts/layer_probe.py 8 800 > $O/layer8/trace.txt 2>&1
python3 scripts/trace_gaps.py $(ls $O/layer8/trace/*kernel_trace.csv | head -1) > $O/layer8/trace_gaps.txt 2>&1
cp $(ls $O/layer8/trace/*kernel_stats.csv | head -1) $O/layer8_kernel_stats.csv
rm -rf $O/layer8/trace
for v in "--variant token_recycle" "--variant eagle2" "--variant eagle2 --model llama3-8b" "--variant eagle" "--model llama3-8b"; do
  tag=$(echo $v | tr -d ' -')
  timeout 500 python3 bench.py $v --no-cpu-baseline --no-live-pmc > $O/bench_$tag.log 2>&1
done
for f in $O/bench_*.log; do echo $f; tail -1 $f | cut -c1-200; done
