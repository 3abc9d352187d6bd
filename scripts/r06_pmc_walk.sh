#!/bin/bash
# r06_pmc_walk.sh <tag> <corpus_tokens> <dist> <slots> -- request / latency / wave counters of k_static_walk in one configuration (one --pmc pass per group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; TOK=$2; DIST=$3; SLOTS=$4
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for c in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -o w -- python3 scripts/walk_probe.py $TOK 1048576 16 3 $DIST $SLOTS > $OUT/p$i.txt 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, json, sys
out = {"tag": sys.argv[2]}
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_static_walk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out[c] = v[-1]
for f in sorted(glob.glob(sys.argv[1] + "/p1/*kernel_trace.csv")):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_static_walk" in r["Kernel_Name"]]
    if d: out["kernel_ns_last"] = d[-1]
if out.get("TCP_TCC_READ_REQ_sum"):
    out["mean_latency_cycles"] = round(out["TCP_TCC_READ_REQ_LATENCY_sum"] / out["TCP_TCC_READ_REQ_sum"], 1)
if out.get("TCC_HIT_sum") is not None:
    out["l2_hit_rate"] = round(out["TCC_HIT_sum"] / max(out["TCC_HIT_sum"] + out["TCC_MISS_sum"], 1), 4)
print(json.dumps(out))
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
