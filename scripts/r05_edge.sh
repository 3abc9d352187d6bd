#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sam.py tests/test_gpu_fullsize.py tests/test_gpu_import.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -12
timeout 1500 python3 scripts/walk_sweep.py gpurun_out/walk_sweep_edge.json --sizes 20,22 --slots 16 > gpurun_out/r05_walk_sweep_edge.log 2>&1; echo "sweep rc $?"
tail -8 gpurun_out/r05_walk_sweep_edge.log
SAMD_EDGE_TABLE=0 timeout 600 python3 scripts/walk_sweep.py gpurun_out/walk_sweep_noedge.json --sizes 20 --slots 16 --no-pmc 2>&1 | tail -4
