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
conv = [k for k in kern if "conv_" in k["kernel"]]
n_conv_plan = sum(1 for l in meta["launches"] if "conv_" in l["kernel"])
out = dict(note="rocprofv3 --kernel-trace --stats of tools/steps_only.py: ResNet50, batch %d, %d steps one batch at a time on one stream with the %s launch plan, nothing else in the process but a clock spin-up of torch matrix products before them (tools/steps_only.py --spinup-ms)" % (meta["batch"], steps, "batches-in-flight (--conc 1: what bench.py's timed region launches)" if meta.get("conc") else "one-batch-at-a-time"),
           conc=meta.get("conc", 0),
           batch=meta["batch"], steps=steps,
           conv_launches_per_step=sum(k["calls_per_step"] for k in conv), conv_launches_per_step_launch_plan=n_conv_plan,
           conv_us_per_step=sum(k["us_per_step"] for k in conv), all_kernels_us_per_step=sum(k["us_per_step"] for k in kern),
           avg_conv_launch_us=sum(k["us_per_step"] for k in conv) / max(1e-9, sum(k["calls_per_step"] for k in conv)),
           kernels=sorted(kern, key=lambda k: -k["us_per_step"]))
assert abs(out["conv_launches_per_step"] - n_conv_plan) < 1e-6, (out["conv_launches_per_step"], n_conv_plan)
json.dump(out, open(out_json, "w"), indent=1)
print("conv kernels: %.1f us per step over %.0f launches (avg %.2f us); all tf2 kernels %.1f us per step" %
      (out["conv_us_per_step"], out["conv_launches_per_step"], out["avg_conv_launch_us"], out["all_kernels_us_per_step"]))
