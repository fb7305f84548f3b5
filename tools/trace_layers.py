#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of tools/steps_only.py -> one row per LAUNCH of the step (median duration over the steps), mapped to
table rows through the library's own launch list (tools/steps_only.py --meta).  Usage: trace_layers.py kernel_trace.csv meta.json [out.json]"""
import csv, json, statistics as st, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tf2::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
meta = json.load(open(sys.argv[2]))
plan = meta["launches"]; n = len(plan); steps = meta["steps"]
assert len(rows) == n * steps, (len(rows), n, steps)
dur = [[] for _ in range(n)]; gap = [[] for _ in range(n)]
for i, r in enumerate(rows):
    k = i % n
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if k: gap[k].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
skip = max(1, steps // 5)                       # the first steps run while the clock still ramps
out = []
tot = 0.0
for k in range(n):
    d = st.median(dur[k][skip:]); g = st.median(gap[k][skip:]) if gap[k] else 0.0
    tot += d
    out.append(dict(index=k, layer=plan[k]["layer"], kernel=plan[k]["kernel"], grid=plan[k]["grid"], us=round(d, 2), gap_before_us=round(g, 2)))
    print(f"{k:3d} row {plan[k]['layer']:3d} {d:8.2f} us  (+{g:5.2f})  grid {plan[k]['grid']:5d}  {plan[k]['kernel']}")
first = [int(rows[s * n]["Start_Timestamp"]) for s in range(skip, steps)]; last = [int(rows[s * n + n - 1]["End_Timestamp"]) for s in range(skip, steps)]
span = st.median([(b - a) / 1e3 for a, b in zip(first, last)])
print(f"sum of kernel medians {tot:.1f} us per step over {n} launches; first start -> last end of a step {span:.1f} us")
if len(sys.argv) > 3:
    json.dump(dict(batch=meta["batch"], steps=steps, conc=meta.get("conc"), launches_per_step=n, kernel_us_per_step=round(tot, 1), step_span_us=round(span, 1), launches=out), open(sys.argv[3], "w"), indent=1)
