#!/bin/bash
# trace + PMC passes for the weight-streaming projections of one decoder layer (scripts/gemm_probe.py); run on the GPU box.
# Counters in their own runs, FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot budget).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o g -- python3 scripts/gemm_probe.py > $OUT/trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$tag -o g -- python3 scripts/gemm_probe.py > $OUT/$tag.txt 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {"counters_per_layer": {}}
for f in glob.glob("gpurun_out/pmc_gemm/*/*counter_collection.csv") + glob.glob("gpurun_out/pmc_gemm/*/*/*counter_collection.csv"):
    agg = collections.defaultdict(float); launches = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "k_gemm_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]] += 1
    for c, v in agg.items():
        out["counters_per_layer"][c] = v / (launches[c] / 4)          # 4 projections per layer
for f in glob.glob("gpurun_out/pmc_gemm/trace/*kernel_stats.csv") + glob.glob("gpurun_out/pmc_gemm/trace/*/*kernel_stats.csv"):
    tot = 0.0; calls = 0
    for r in csv.DictReader(open(f)):
        if "k_gemm_" in r["Name"] and "pack" not in r["Name"]:
            tot += float(r["TotalDurationNs"]); calls += int(r["Calls"])
            out.setdefault("kernels", []).append({k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")})
    out["ns_per_layer"] = tot / (calls / 4)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_gemm/summary.json", "w"), indent=1)
PY
tail -1 $OUT/trace.txt
rm -rf $OUT/trace $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/TCC_*
