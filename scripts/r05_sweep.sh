#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python3 scripts/walk_sweep.py gpurun_out/walk_sweep.json > gpurun_out/r05_walk_sweep.log 2>&1; echo "sweep rc $?"
tail -20 gpurun_out/r05_walk_sweep.log
timeout 600 python -m pytest tests/test_gpu_lm_shapes.py -m gpu -x -q -k "vicuna_7b" -s 2>&1 | tail -8
