#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
for v in "" "-DPA_MINW8=3 -DPA_MINW4=1" "-DPA_MINW8=2 -DPA_MINW4=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $v -o build/pa_v scripts/probes/prefill_attn_probe.hip || exit 1
  echo "== variant [$v]"; for wv in 4 8; do for a in "512 32 32 0" "1024 32 32 0" "1536 32 32 0" "1 32 32 2000"; do timeout 100 build/pa_v $a 20 2048 $wv | tr '\n' ' '; echo; done; done
done
