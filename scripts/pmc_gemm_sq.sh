#!/bin/bash
# SQ / LDS / L2 counters of the weight-streaming projections at two row tiles (scripts/gemm_probe.py 24 <rows>); run on the GPU box.
# usage: scripts/pmc_gemm_sq.sh "16 64"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm_sq
rm -rf $OUT; mkdir -p $OUT
for R in ${1:-16 64}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$R -o g -- python3 scripts/gemm_probe.py 24 $R > $OUT/trace_$R.txt 2>&1
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p${R}_$i -o g -- python3 scripts/gemm_probe.py 24 $R > $OUT/p${R}_$i.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, json, re
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_gemm_sq/p*_*")):
    R = re.search(r"/p(\d+)_", d).group(1)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_gemm_skinny" in r["Kernel_Name"]:
                silu = ", 1>" in r["Kernel_Name"]
                k = (r["Counter_Name"], "silu" if silu else "split")
                agg[k] += float(r["Counter_Value"]); n[k] += 1
        for (c, kind), v in agg.items():
            out.setdefault(f"rows{R}", {}).setdefault(kind, {})[c] = v / n[(c, kind)]
for d in sorted(glob.glob("gpurun_out/pmc_gemm_sq/trace_*")):
    if not d.endswith(".txt"):
        R = d.rsplit("_", 1)[1]
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_gemm_skinny" in r["Name"]:
                    out.setdefault(f"rows{R}", {}).setdefault("kernels", []).append({k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs")})
json.dump(out, open("gpurun_out/pmc_gemm_sq/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
