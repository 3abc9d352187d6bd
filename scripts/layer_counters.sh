#!/bin/bash
# Per-kernel durations, launch gaps and wave counters of the verify forward at one row bucket (run on the GPU box):
#   scripts/layer_counters.sh [tag] [rows] [L]      -> gpurun_out/<tag>/{trace_gaps.txt, counters.json}
# Counters are collected in their own passes (never together with --sys-trace etc.).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04_layer}; R=${2:-8}; L=${3:-800}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 scripts/layer_probe.py $R $L > $OUT/trace.txt 2>&1
python3 scripts/trace_gaps.py $(ls $OUT/trace/*kernel_trace.csv | head -1) > $OUT/trace_gaps.txt 2>&1
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag -o p -- python3 scripts/layer_probe.py $R $L 3 > $OUT/pmc_$tag.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, open(d + "/counters.json", "w"), indent=1)
for k, cs in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(k, {c: round(x, 1) for c, x in cs.items()})
PY
