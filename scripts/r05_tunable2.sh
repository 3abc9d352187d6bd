#!/bin/bash
# TunableOp with cold operands (rotating buffers) and longer timing, on the prefill's shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=1024 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=50 PYTORCH_TUNABLEOP_VERBOSE=0
timeout 300 python3 scripts/gemm_shapes_probe.py 512 1280 > gpurun_out/r05_gemm_untuned2.log 2>&1; grep "^M=" gpurun_out/r05_gemm_untuned2.log
SAMD_PROBE_TUNABLE=gpurun_out/r05_tunableop2.csv SAMD_PROBE_TUNABLE_ENV=1 timeout 2400 python3 scripts/gemm_shapes_probe.py 512 1280 > gpurun_out/r05_gemm_tuned2.log 2>&1; grep "^M=" gpurun_out/r05_gemm_tuned2.log
cat gpurun_out/r05_tunableop2.csv | cut -c1-160
