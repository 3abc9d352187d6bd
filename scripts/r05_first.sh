#!/bin/bash
# round 5, first GPU pass: the GPU suite, the driver's bench command, the summarization leg at 24 requests per profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gputest.log; tail -5 gpurun_out/r05_gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver.log 2> gpurun_out/r05_bench_driver.err; echo "bench rc $?"
timeout 900 python bench.py --workload summarization --steps 600 --no-long-run > gpurun_out/r05_bench_summarization.log 2> gpurun_out/r05_bench_summarization.err; echo "bench summ rc $?"
python scripts/show_line.py gpurun_out/r05_bench_summarization.log value ms_per_step speedup_vs_ar step_breakdown_by_rows summarization roofline | head -230
