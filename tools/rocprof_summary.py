#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSV of tools/steps_only.py -> per-step figures: the conv kernels' time per step, per-kernel
average duration, and the library's own launch list beside it (tools/steps_only.py --meta).  bench.py reads the result
(profiles/rNN_rocprof_b<batch>_summary.json) into roofline.kernel_us_per_step_rocprof."""
import csv, json, sys
stats_csv, meta_json, out_json = sys.argv[1:4]
meta = json.load(open(meta_json))
steps = meta["steps"]
rows = list(csv.DictReader(open(stats_csv)))
kern = []
for r in rows:
    name = r["Name"]
    if not name.startswith("tf2::") and "tf2::" not in name:
        continue
    kern.append(dict(kernel=name.split("(")[0], calls=int(r["Calls"]), calls_per_step=int(r["Calls"]) / steps,
                     total_us=float(r["TotalDurationNs"]) / 1e3, average_us=float(r["AverageNs"]) / 1e3,
                     us_per_step=float(r["TotalDurationNs"]) / 1e3 / steps))
# launches of CONVOLUTION rows (the row's conv kernel(s) + its own pool / average launch: what bench.py's per-row HIP events bracket); the
# input preparation, L2Norm and independent-pool rows are not.  A kernel name that serves both kinds of rows (maxpool_kernel) counts by
# the share of its launches that belong to convolution rows.
conv_rows = set(meta.get("conv_rows", [l["layer"] for l in meta["launches"] if l["layer"] >= 0]))
base = lambda n: n.replace("void ", "").replace("tf2::", "").split("<")[0].split("(")[0].strip()
n_all, n_cv = {}, {}
for l in meta["launches"]:
    b = base(l["kernel"])
    n_all[b] = n_all.get(b, 0) + 1
    n_cv[b] = n_cv.get(b, 0) + (1 if l["layer"] in conv_rows else 0)
for k in kern:
    b = base(k["kernel"])
    k["conv_row_share"] = n_cv.get(b, 0) / n_all[b] if n_all.get(b) else 0.0
conv = [dict(k, us_per_step=k["us_per_step"] * k["conv_row_share"], calls_per_step=k["calls_per_step"] * k["conv_row_share"]) for k in kern if k["conv_row_share"] > 0]
n_conv_plan = sum(1 for l in meta["launches"] if l["layer"] in conv_rows)
out = dict(note="rocprofv3 --kernel-trace --stats of tools/steps_only.py: %s, batch %d, %d steps one batch at a time on one stream with the %s launch plan, nothing else in the process but a clock spin-up of torch matrix products before them (tools/steps_only.py --spinup-ms)" % (meta.get("net", "ResNet50"), meta["batch"], steps, "batches-in-flight (--conc 1: what bench.py's timed region launches)" if meta.get("conc") else "one-batch-at-a-time"),
           conc=meta.get("conc", 0), net=meta.get("net", "ResNet50"),
           batch=meta["batch"], steps=steps,
           conv_launches_per_step=sum(k["calls_per_step"] for k in conv), conv_launches_per_step_launch_plan=n_conv_plan,
           conv_us_per_step=sum(k["us_per_step"] for k in conv), all_kernels_us_per_step=sum(k["us_per_step"] for k in kern),
           avg_conv_launch_us=sum(k["us_per_step"] for k in conv) / max(1e-9, sum(k["calls_per_step"] for k in conv)),
           kernels=sorted(kern, key=lambda k: -k["us_per_step"]))
assert abs(out["conv_launches_per_step"] - n_conv_plan) < 1e-6, (out["conv_launches_per_step"], n_conv_plan)
json.dump(out, open(out_json, "w"), indent=1)
print("conv kernels: %.1f us per step over %.0f launches (avg %.2f us); all tf2 kernels %.1f us per step" %
      (out["conv_us_per_step"], out["conv_launches_per_step"], out["avg_conv_launch_us"], out["all_kernels_us_per_step"]))
