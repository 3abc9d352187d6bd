#!/bin/bash
# everything profiles/r05_wide_gemm.md quotes, on one box: the hand-written prefill GEMM (whole tiles / stream-K, both partial-tile protocols,
# the SiLU epilogue) and the library on the same shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Iscripts/probes"
$H -o build/wg scripts/probes/wide_gemm_probe.hip || exit 1
$H -DWIDE_SC1=0 -o build/wg_fence scripts/probes/wide_gemm_probe.hip || exit 1
$H -DWIDE_KSTEP=0 -o build/wg_k0 scripts/probes/wide_gemm_probe.hip || exit 1
LAYER="22016 4096|12288 4096|4096 4096|4096 11008"
{
echo "== A. whole tiles (mode 0): rounds, k scaling, one tile per XCD pair"
for s in "1024 16384 2048" "1024 16384 4096" "1024 16384 8192" "2048 16384 4096" "256 4096 2048" "256 4096 8192"; do timeout 120 build/wg $s 0 20 1; done
echo "== A2. the same, every k-tile re-reading the first (nothing beyond the L2)"
for s in "1024 16384 4096" "1024 16384 8192"; do timeout 120 build/wg_k0 $s 0 20 0; done
echo "== B. stream-K (mode 1), partial tiles through sc1 stores / loads"
for M in 512 1024 1536; do IFS='|' read -ra L <<< "$LAYER"; for nk in "${L[@]}"; do timeout 120 build/wg $M $nk 1 20 1; done; done
timeout 120 build/wg 1024 16384 4096 1 20 1
echo "== C. stream-K, partial tiles through plain stores + agent-scope release / acquire fences"
for s in "512 22016 4096" "1536 22016 4096" "1536 12288 4096" "1024 16384 4096"; do timeout 120 build/wg_fence $s 1 20 1; done
echo "== D. gate|up with the SiLU * up epilogue"
for M in 512 1024 1536; do timeout 120 build/wg $M 22016 4096 1 20 1 1; done
echo "== E. whole tiles on the layer's shapes"
for M in 512 1536; do IFS='|' read -ra L <<< "$LAYER"; for nk in "${L[@]}"; do timeout 120 build/wg $M $nk 0 20 0; done; done
echo "== F. the library (torch.mm -> hipBLASLt) on the same box"
timeout 300 python3 scripts/gemm_shapes_probe.py 512 1024 1536
} > gpurun_out/r05_wide_gemm.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r05_wide_gemm.log
