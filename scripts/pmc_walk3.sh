#!/bin/bash
#   scripts/pmc_walk3.sh [tag] [first pass]   every pass under its own 90 s timeout: an unsupported counter group made rocprofv3 abort and then hang
# r04: what the SAM traversal kernel waits for -- address translation, texture-path and L2 counters (one --pmc pass per group; run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc_walk3}; mkdir -p $OUT
i=0
for c in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
         "TCC_BUSY_sum TCC_REQ_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  if [ $i -lt ${2:-1} ]; then continue; fi
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
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
