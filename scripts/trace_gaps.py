"""trace_gaps.py <kernel_trace.csv> -- per-kernel durations and the idle gaps between consecutive kernels of the verify
forward (rocprofv3 --kernel-trace --output-format csv), to see what fusing two launches could save."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = lambda r: r["Kernel_Name"].split("(")[0][:40]
# keep the last 40 % of the trace (steady-state graph replays)
rows = rows[int(len(rows) * 0.6):]
dur = collections.defaultdict(list); gap_after = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[names(a)].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000:
        gap_after[names(a) + " -> " + names(b)].append(g)
print("durations (us):")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:42s} n={len(v):6d} mean={sum(v)/len(v)/1e3:7.2f} total_ms={sum(v)/1e6:8.2f}")
print("gaps (us) between consecutive kernels:")
tot = 0
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {k:86s} n={len(v):6d} mean={sum(v)/len(v)/1e3:6.2f}")
allg = [g for v in gap_after.values() for g in v]
alld = [d for v in dur.values() for d in v]
print(f"sum of kernel time {sum(alld)/1e6:.2f} ms, sum of gaps {sum(allg)/1e6:.2f} ms over {len(alld)} kernels (mean gap {sum(allg)/len(allg)/1e3:.2f} us)")
