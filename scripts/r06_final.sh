#!/bin/bash
# round-6 closing measurements at HEAD (GPU box).  Every step under its own timeout; core dumps off.
#   A: the GPU suite (plain + a subset on a NaN-poisoned allocator; T = only these), the traversal-kernel sweep (12 rows), its counters
#   B: bench lines -- the driver's window, the 2000-step default, the summarization leg at 24 requests, rocprofv3 kernel stats, variants
#   C: per-kernel tables of the 8- and 64-row layers, the attention pair, the EAGLE-2 draft
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
O=gpurun_out/final6; mkdir -p $O
what=${1:-all}
if [ "$what" = all ] || [ "$what" = A ] || [ "$what" = T ]; then
  timeout 900 python3 -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log; tail -3 $O/gputest.log
  SAMD_TEST_POISON=1 SAMD_TEST_POISON_GIB=24 timeout 900 python3 -m pytest tests/test_gpu_wide_drafts.py tests/test_gpu_sam.py tests/test_gpu_api.py tests/test_gpu_llama.py tests/test_gpu_verify.py tests/test_gpu_prefill_shaping.py tests/test_gpu_vt_cache.py -m gpu -q > $O/gputest_poison.log 2>&1; echo "pytest rc $?" >> $O/gputest_poison.log; tail -3 $O/gputest_poison.log
fi
if [ "$what" = all ] || [ "$what" = A ]; then
  timeout 2700 python3 scripts/walk_sweep.py $O/walk_sweep.json > $O/walk_sweep.log 2>&1; echo "sweep rc $?"; tail -14 $O/walk_sweep.log
  bash scripts/pmc_walk.sh final6/walk_pmc > $O/pmc_walk.log 2>&1; tail -2 $O/pmc_walk.log | cut -c1-400
fi
if [ "$what" = all ] || [ "$what" = B ]; then
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_20steps.log 2> $O/bench_driver_20steps.err
  timeout 600 python3 bench.py > $O/bench_default.log 2> $O/bench_default.err
  timeout 600 python3 bench.py --no-cpu-baseline --no-live-pmc --summ-requests 0 > $O/bench_default_rep1.log 2>&1
  timeout 900 python3 bench.py --workload summarization --steps 600 --no-long-run > $O/bench_summarization.log 2> $O/bench_summarization.err
  timeout 700 scripts/bench_stats.sh final6_stats --no-cpu-baseline --no-live-pmc --summ-requests 0 > $O/bench_stats.txt 2>&1
  for v in "--variant token_recycle" "--variant eagle2 --model llama3-8b" "--model llama3-8b"; do
    tag=$(echo $v | tr -d ' -')
    timeout 600 python3 bench.py $v --no-cpu-baseline --no-live-pmc > $O/bench_$tag.log 2>&1
  done
  for f in $O/bench_*.log; do echo $f; python3 scripts/show_line.py $f value ms_per_step mean_accepted_tokens speedup_vs_ar 2>/dev/null | tr '\n' ' '; echo; done
fi
if [ "$what" = all ] || [ "$what" = C ]; then
  for R in 8 64; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/layer$R -o t -- python3 scripts/layer_probe.py $R 800 > $O/layer$R.txt 2>&1
    cp $(ls $O/layer$R/*kernel_stats.csv | head -1) $O/layer${R}_kernel_stats.csv
    python3 scripts/show_stats.py $O/layer${R}_kernel_stats.csv | head -14
  done
  for i in 1 2; do python3 scripts/attn_ab.py vt 800 300 1500; python3 scripts/attn_ab.py vt 800 gqa; done > $O/attn_ab.log 2>&1; cat $O/attn_ab.log
  python3 scripts/gemm_groups_ab.py > $O/gemm_groups_ab.log 2>&1; cat $O/gemm_groups_ab.log
fi
