#!/bin/bash
# r04: request and latency counters of the SAM traversal kernel (one --pmc pass per group, each under its own timeout; run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc_walk4}; mkdir -p $OUT
i=0
for c in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -o w -- python3 scripts/walk_probe.py 4194304 1048576 16 3 > $OUT/p$i.txt 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_static_walk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out[c] = v[-1]
print(json.dumps(out))
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
