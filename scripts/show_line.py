"""show_line.py FILE [key ...] -- pretty-print (parts of) the JSON line a bench.py run wrote to FILE"""
import json, sys
d = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{") and '"metric"' in line:
        d = json.loads(line)
if d is None:
    print("no bench line in", sys.argv[1]); print(open(sys.argv[1]).read()[-1500:]); sys.exit(1)
keys = sys.argv[2:] or ["value", "ms_per_step", "mean_accepted_tokens", "speedup_vs_ar", "step_breakdown_by_rows"]
for k in keys:
    print(k, "=", json.dumps(d.get(k), indent=1))
