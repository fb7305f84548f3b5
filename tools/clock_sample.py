#!/usr/bin/env python3
"""Engine clock and power while bench.py runs (rocm-smi sampled every ~50 ms in a side process): does the chip hold its clock with
four batches in flight?  usage: clock_sample.py [bench.py arguments...]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["--no-cpu", "--steps", "3000", "--warmup", "10", "--extra-batches", ""]
p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
t0 = time.time()
while p.poll() is None:
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
    except Exception as e:      # noqa: BLE001
        out = str(e)
    sclk = re.search(r"sclk clock level:\s*\d*:?\s*\(?(\d+)Mhz", out)
    pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
    samples.append((round(time.time() - t0, 2), int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))
    time.sleep(0.05)
line = p.stdout.read().strip().splitlines()[-1]
print("samples (s, sclk MHz, W):", samples[::max(1, len(samples) // 40)])
if not any(s[1] for s in samples):
    print(out[:1500])
import json
d = json.loads(line)
print("bench:", d["value"], "img/s,", d["ms_per_step"], "ms per step; one batch at a time", d.get("images_per_s_one_batch_at_a_time"))
