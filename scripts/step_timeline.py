"""step_timeline.py <kernel_trace.csv> [min_kernels_per_step] -- GPU idle time inside decode steps of a bench run (rocprofv3
--kernel-trace).  A step starts at the verify forward's first kernel (k_embed_rows); steps with at least `min_kernels` kernels are
kept (drops prefill chunks / autoregressive steps when looking at a plugin run).  Prints the median step's span, busy and idle
time and the gaps (> 3 us) of that step with the kernels around them."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
min_k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
starts = [i for i, r in enumerate(rows) if "k_embed_rows" in r["Kernel_Name"]]
steps = []
for a, b in zip(starts, starts[1:]):
    if b - a >= min_k:
        ks = rows[a:b]
        span = int(rows[b]["Start_Timestamp"]) - int(ks[0]["Start_Timestamp"])
        busy = sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in ks)
        steps.append((span, busy, a, b))
if not steps:
    sys.exit("no steps found")
steps.sort()
span, busy, a, b = steps[len(steps) // 2]
print(f"{len(steps)} steps with >= {min_k} kernels; median step: span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us, {b-a} kernels")
print(f"mean span {statistics.mean(s[0] for s in steps)/1e3:.1f} us, mean idle {statistics.mean(s[0]-s[1] for s in steps)/1e3:.1f} us")
ks = rows[a:b + 1]
for x, y in zip(ks, ks[1:]):
    g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
    if g > 3000:
        print(f"  gap {g/1e3:7.1f} us at +{(int(x['End_Timestamp'])-int(ks[0]['Start_Timestamp']))/1e3:8.1f} us   {name(x)}  ->  {name(y)}")
