#!/bin/bash
# PMC + trace passes for the SAM traversal kernel at bench.py's roofline configuration (run on the GPU box):
#   scripts/pmc_walk.sh [tag]        tag = output sub-directory under gpurun_out/ (default pmc_r2); SAMD_WALK_CHAIN=0 in the
#                                    environment profiles the node-only variant (the A/B of profiles/r02_walk_pmc.md)
# Counters are collected in their own runs (never together with --sys-trace etc.); FETCH_SIZE and WRITE_SIZE in separate passes.
# The summary (walk_summary.json) is what profiles/walk_pmc.json -- bench.py's `roofline.traffic` -- is refreshed from.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc_r2}
mkdir -p $OUT
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/walk_trace -o w -- python3 scripts/walk_probe.py 4194304 1048576 16 8 > $OUT/walk_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $c | tr ' ' '+')
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/walk_$tag -o w -- python3 scripts/walk_probe.py > $OUT/walk_$tag.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out, d = {}, sys.argv[1]
for f in glob.glob(d + "/walk_*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_static_walk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out[c] = v[-1]
for f in glob.glob(d + "/walk_trace/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_static_walk" in r["Name"]:
            out["kernel_stats"] = {k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")}
# ONE averaging convention for this kernel's duration: WARM launches -- every launch of the probe but the first (cold
# caches, first touch of the tables) -- from the kernel trace; bench.py's HIP-event figure is the mean of 20 launches after a warm-up one
for f in glob.glob(d + "/walk_trace/*kernel_trace.csv"):
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_static_walk" in r["Kernel_Name"]]
    if len(dur) > 1:
        out["warm_kernel_ns"] = {"launches": len(dur) - 1, "mean": sum(dur[1:]) / (len(dur) - 1), "min": min(dur[1:]), "cold_first": dur[0]}
for line in open(d + "/walk_trace.txt"):
    if line.startswith("{'bound'"):
        out["bench_roofline"] = eval(line)
# the sources the counters belong to: bench.py nulls `roofline.traffic` when the tree's kernel sources hash differently
import hashlib, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
h = hashlib.sha256()
for name in ("sam_kernels.hip", "sam_device.h", "samd_common.h"):
    h.update(open(os.path.join(root, "sam-decoding_amd", "csrc", name), "rb").read())
out["kernel_source_sha16"] = h.hexdigest()[:16]
print(json.dumps(out))
json.dump(out, open(d + "/walk_summary.json", "w"), indent=1)
PY
