"""request_timeline.py <kernel_trace.csv> -- where the wall time of a bench run goes BETWEEN decode steps (rocprofv3 --kernel-trace):
the host turnaround after every step (last kernel of a step -> first kernel of the next: report D2H, host wake-up, bucket choice,
hipGraphLaunch) and the start of a request (last decode kernel of one request -> first decode kernel of the next: prefill, prompt
ingest, first draft), split into GPU-busy and idle time.  A decode step starts at k_embed_rows and holds k_session; a prefill
holds library GEMMs (Cijk_*)."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
starts = [i for i, r in enumerate(rows) if "k_embed_rows" in r["Kernel_Name"]]
segs = []                                            # (first, last+1, kind)
for a, b in zip(starts, starts[1:] + [len(rows)]):
    names = [r["Kernel_Name"] for r in rows[a:b]]
    kind = "prefill" if any(n.startswith("Cijk") for n in names) else ("step" if any("k_session" in n for n in names) else "other")
    segs.append((a, b, kind))
turn, req = [], []
for (a, b, k), (a2, b2, k2) in zip(segs, segs[1:]):
    if k == "step" and k2 == "step":
        last = max(E(r) for r in rows[a:b])
        turn.append((S(rows[a2]) - last) / 1e3)
i = 0
while i < len(segs):
    if segs[i][2] == "prefill":
        j = i
        while j < len(segs) and segs[j][2] != "step":
            j += 1
        if j < len(segs) and i > 0 and segs[i - 1][2] == "step":
            t0 = max(E(r) for r in rows[segs[i - 1][0]:segs[i - 1][1]])
            t1 = S(rows[segs[j][0]])
            busy = sum(E(r) - S(r) for r in rows[segs[i][0]:segs[j][0]])
            req.append(((t1 - t0) / 1e3, busy / 1e3))
        i = j
    i += 1
print(f"{len(turn)} step->step turnarounds: median {statistics.median(turn):.1f} us, mean {statistics.mean(turn):.1f} us, p90 {sorted(turn)[int(len(turn)*0.9)]:.1f} us")
if req:
    print(f"{len(req)} request starts: wall median {statistics.median(r[0] for r in req)/1e3:.2f} ms, GPU-busy (sum of kernels, streams overlap) median {statistics.median(r[1] for r in req)/1e3:.2f} ms")
    for w, bz in req[:6]:
        print(f"   wall {w/1e3:.2f} ms  busy {bz/1e3:.2f} ms")
