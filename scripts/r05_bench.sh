#!/bin/bash
# round 5: refresh profiles/walk_pmc.json at HEAD, the driver's bench command, the 2000-step default run, the summarization run, kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/pmc_walk.sh r05_walk > gpurun_out/r05_pmc_walk.log 2>&1; tail -3 gpurun_out/r05_pmc_walk.log | cut -c1-600
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver.log 2> gpurun_out/r05_bench_driver.err; echo "driver rc $?"
timeout 900 python bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err; echo "default rc $?"
timeout 900 python bench.py --workload summarization --steps 600 --no-long-run > gpurun_out/r05_bench_summarization.log 2> gpurun_out/r05_bench_summarization.err; echo "summ rc $?"
for f in driver default summarization; do python scripts/show_line.py gpurun_out/r05_bench_$f.log value ms_per_step mean_accepted_tokens speedup_vs_ar | tr '\n' ' '; echo; done
python scripts/show_line.py gpurun_out/r05_bench_default.log step_breakdown_by_rows roofline | head -80
