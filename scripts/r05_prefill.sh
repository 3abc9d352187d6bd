#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lm_shapes.py -m gpu -x -q -k "vicuna_7b" -s 2>&1 | tail -6
timeout 300 python3 scripts/gemm_shapes_probe.py > gpurun_out/r05_gemm_shapes.log 2>&1; cat gpurun_out/r05_gemm_shapes.log
for n in 512 1536; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prefill_$n -o p -- python3 scripts/prefill_probe.py $n > gpurun_out/r05_prefill_$n.log 2>&1
tail -2 gpurun_out/r05_prefill_$n.log
python3 scripts/show_stats.py $(find gpurun_out/prefill_$n -name "*kernel_stats.csv" | head -1) 22
done
