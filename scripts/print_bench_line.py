import json,sys
l=sys.stdin.read()
try:
    d=json.loads(l); print(d["config"]["variant"], d["config"]["model_shape"], d["value"], d["ms_per_step"], d["mean_accepted_tokens"], {k:v["step_ms"] for k,v in d["step_breakdown_by_rows"].items()})
except Exception as e: print("ERR", l[-800:])
