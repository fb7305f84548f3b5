#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and reports kernel durations and the gaps between consecutive kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
dur = collections.defaultdict(list); gaps = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[r["Kernel_Name"][:60]].append(e - s)
    if prev_end is not None: gaps.append(s - prev_end)
    prev_end = e
tot = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(sum(v) for v in dur.values())
import statistics as st
print(f"kernels {len(rows)} span {tot/1e3:.1f} us busy {busy/1e3:.1f} us ({100*busy/tot:.1f}%)")
g = sorted(gaps)
print(f"gaps: median {st.median(g)/1e3:.2f} us mean {st.mean(g)/1e3:.2f} us p10 {g[len(g)//10]/1e3:.2f} p90 {g[9*len(g)//10]/1e3:.2f} max {g[-1]/1e3:.1f}")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/1e3:10.1f} us total {len(v):6d} calls {st.mean(v)/1e3:8.2f} us avg  {k}")
