#!/bin/bash
# the library GEMMs of the prefill over row counts (untuned), then with TunableOp's exhaustive pick on the row counts the bench uses
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 400 python3 scripts/gemm_shapes_probe.py 512 640 768 896 1024 1152 1280 1408 1536 1664 > gpurun_out/r05_gemm_rows.log 2>&1; cat gpurun_out/r05_gemm_rows.log
SAMD_PROBE_TUNABLE=gpurun_out/r05_tunableop.csv timeout 1200 python3 scripts/gemm_shapes_probe.py 512 1024 1280 1536 > gpurun_out/r05_gemm_tuned.log 2>&1; tail -5 gpurun_out/r05_gemm_tuned.log
wc -l gpurun_out/r05_tunableop*.csv
