#!/bin/bash
# where does the 64-row gate|up launch lose its time?  (diagnostic library, see scripts/pairs_ablate.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/pairs_ablate.py 16 32 64
for abl in 0 1 2 3 4 5 6 7 8 9 10 11; do
  SAMD_HIP_LIB=$PWD/scripts/ab/libsamd_hip_abl.so SAMD_GEMM_ABL=$abl timeout 120 python scripts/pairs_ablate.py 16 32 64
done
} > gpurun_out/r05_pairs_ablate.log 2>&1
cat gpurun_out/r05_pairs_ablate.log | grep ABL
