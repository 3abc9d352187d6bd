#!/bin/bash
# PMC + trace passes for the SAM traversal kernel at bench.py's roofline configuration (run on the GPU box).
# Counters are collected in their own runs; FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot budget).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_r1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/walk_trace -o w -- python3 scripts/walk_probe.py > $OUT/walk_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/walk_$tag -o w -- python3 scripts/walk_probe.py > $OUT/walk_$tag.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/pmc_r1/walk_*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_static_walk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out[c] = v[-1]
for f in glob.glob("gpurun_out/pmc_r1/walk_trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_static_walk" in r["Name"]:
            out["kernel_stats"] = {k: r[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs")}
print(json.dumps(out))
json.dump(out, open("gpurun_out/pmc_r1/walk_summary.json", "w"), indent=1)
PY
tail -1 $OUT/walk_trace.txt
