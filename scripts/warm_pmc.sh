#!/bin/bash
# do the projections hit what the glue launches warmed?  TCC hit / miss and fetched bytes per kernel, warm-up off vs on; GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
for kb in "$@"; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum"; do
    tag=$(echo $c | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/wpmc_${kb}_$tag -o w -- python3 scripts/warm_sweep.py $kb > /tmp/wpmc_$kb.log 2>&1
    python3 - /tmp/wpmc_${kb}_$tag $kb <<'PY'
import csv, glob, sys, collections
f = (glob.glob(sys.argv[1] + "/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if any(s in k for s in ("k_gemm_skinny", "k_rmsnorm", "k_attn_combine", "k_tree_attention")):
        print(f"warm {sys.argv[2]:>3s} KiB  {k:60s} {c:22s} per launch {v / n:12.0f}  ({n} launches)")
PY
  done
done
