#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and reports how much the kernels of concurrent HIP streams overlap on the GPU:
sum of kernel durations, union of their [start, end] intervals, time with >= 2 (>= 3) kernels resident, per stream/queue
counts.  Evidence for bench.py's batches-in-flight figure (the serial kernel time of a step exceeds ms_per_step only
because kernels of different batches run concurrently)."""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip_frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5      # keep the last part of the run (steady state)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "tf2::" in r["Kernel_Name"] or "copyBuffer" in r["Kernel_Name"]]
rows = rows[int(len(rows) * skip_frac):]
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth = 0; last = ev[0][0]; at = collections.Counter()
for t, d in ev:
    at[depth] += t - last
    last = t; depth += d
span = ev[-1][0] - ev[0][0]
total = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
busy = span - at[0]
queues = collections.Counter(r.get("Queue_Id", "?") for r in rows)
out = dict(kernels=len(rows), span_us=round(span / 1e3, 1), sum_of_kernel_durations_us=round(total / 1e3, 1),
           union_busy_us=round(busy / 1e3, 1), overlap_factor=round(total / busy, 3),
           frac_time_ge2_kernels=round(sum(v for k, v in at.items() if k >= 2) / span, 3),
           frac_time_ge3_kernels=round(sum(v for k, v in at.items() if k >= 3) / span, 3),
           frac_time_idle=round(at[0] / span, 3), kernels_per_queue=dict(queues))
print(json.dumps(out))
