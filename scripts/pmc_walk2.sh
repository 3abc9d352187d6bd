#!/bin/bash
# usage: pmc_walk2.sh <counters...>  -- one rocprofv3 PMC pass over scripts/walk_probe.py, per-kernel averages (both walk kernels); GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=/tmp/pmc_w2_$$; mkdir -p gpurun_out/r03
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o w -- python3 scripts/walk_probe.py > $OUT.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = (glob.glob(sys.argv[1] + "/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_static_walk" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:42s} {c:24s} per launch (last of {len(v)}): {v[-1]:.0f}")
PY
