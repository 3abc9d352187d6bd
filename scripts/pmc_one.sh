#!/bin/bash
# usage: pmc_one.sh <tag> <counters...>  -- one rocprofv3 PMC pass over scripts/walk_probe.py (k_static_walk rows only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/pmc_r1/$tag; mkdir -p gpurun_out/pmc_r1
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o w -- python3 scripts/walk_probe.py > $OUT.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_static_walk" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print(f"{c:28s} per launch (last of {len(v)}): {v[-1]:.0f}")
PY
