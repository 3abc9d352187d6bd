#!/bin/bash
# the per-CU ingest question of the 64-row tile: probe + counters of the shipped gate|up launch at 8 / 64 rows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o build/ingest_probe scripts/probes/ingest_probe.hip || exit 1
{
for depth in 2 3; do
  for args in "0 0" "4 32" "4 8" "1 32" "5 32" "2 32" "3 32" "1 8" "1 16"; do
    timeout 60 build/ingest_probe $args 11 $depth 30
  done
done
} > gpurun_out/r05_ingest_probe.log 2>&1
cat gpurun_out/r05_ingest_probe.log
for rows in 8 64; do
  for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_sum TA_BUSY_max" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    tag=$(echo $c | tr ' ' '+')
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/ingest_pmc/r${rows}_$tag -o p -- python3 scripts/layer_probe.py $rows 800 3 > gpurun_out/ingest_pmc_r${rows}_$tag.txt 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for rows in (8, 64):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/ingest_pmc/r{rows}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[rows] = {k: {c: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for c, v in cs.items()} for k, cs in agg.items() if "gemm" in k}
json.dump(out, open("gpurun_out/r05_ingest_counters.json", "w"), indent=1)
for rows, ks in out.items():
    for k, cs in ks.items():
        print(rows, k[:60], {c: round(x, 1) for c, x in cs.items()})
PY
