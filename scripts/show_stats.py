"""print a compact view of a rocprofv3 *_kernel_stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f'{r["Name"][:72]:72s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:8.2f} min={float(r["MinNs"])/1e3:7.2f} max={float(r["MaxNs"])/1e3:8.2f} pct={float(r["Percentage"]):6.2f}')
