#!/usr/bin/env python3
"""Where the wall time of one render goes, from a rocprofv3 kernel trace (CSV): time with no kernel running, and time by
the set of kernel kinds running at once.  python tools/step_timeline.py <..._kernel_trace.csv>  (looks at the last render:
from the last k_raygen to the last k_finalize)"""
import csv, sys
KIND = (("k_extend", "E"), ("k_shade", "S"), ("k_finish", "F"), ("k_accumulate", "A"), ("k_raygen", "G"), ("k_finalize", "Z"))
def kind(n):
    for k, c in KIND:
        if k in n: return c
    return "o"
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"])))
ev.sort()
t0 = max(s for s, e, k in ev if k == "G")
t1 = max(e for s, e, k in ev if k == "Z")
ev = [x for x in ev if x[0] >= t0 and x[1] <= t1]
pts = []
for s, e, k in ev:
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
cnt, last, state = {}, t0, {}
for t, d, k in pts:
    key = "".join(sorted(c for c, n in cnt.items() if n > 0)) or "idle"
    state[key] = state.get(key, 0) + t - last
    cnt[k] = cnt.get(k, 0) + d; last = t
tot = t1 - t0
print(f"render {tot / 1e6:.2f} ms, {len(ev)} kernels")
for k, v in sorted(state.items(), key=lambda kv: -kv[1]):
    print(f"  {k:8s} {v / 1e6:8.3f} ms  {100.0 * v / tot:5.1f} %")
gaps = []
last_end = t0
for s, e, k in ev:
    if s > last_end: gaps.append(s - last_end)
    last_end = max(last_end, e)
gaps.sort(reverse=True)
print("  idle gaps:", len(gaps), "largest (us):", [round(g / 1e3, 1) for g in gaps[:12]])
