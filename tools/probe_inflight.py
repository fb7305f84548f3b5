#!/usr/bin/env python3
"""Marginal in-flight cost of every launch group (probe build, skip_layers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TF2_AMD_LIB"] = os.path.join(ROOT, "tf2_amd", "libtf2amd_probe.so"); os.environ["TF2_AMD_TOOL_LIB"] = "1"
from tf2_amd._lib import set_opts  # noqa: E402
import numpy as np, torch
from tf2_amd import config as cfg, network, synth, _lib
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, 32, 1)).to("cuda:0")
streams = [torch.cuda.Stream(device="cuda:0") for _ in range(4)]
def inflight(n=4, steps=120):
    rs = [network.Runner(None, net) for _ in range(n)]
    for k in range(4 * n):
        with torch.cuda.stream(streams[k % n]): rs[k % n].run_batch(x, concurrency=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % n]): rs[k % n].run_batch(x, concurrency=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
# spin up
for _ in range(3): inflight()
groups = ["0-0", "1-2", "3-4", "5-5", "6-7", "8-8", "9-10", "11-12", "13-13", "14-14", "15-17", "18-20", "21-23", "24-25", "26-26", "27-27",
          "28-30", "31-33", "34-36", "37-39", "40-42", "43-44", "45-45", "46-46", "47-47", "48-48", "49-49", "50-50", "51-51", "52-52", "53-53",
          "0-10", "11-14", "15-23", "24-27", "28-42", "43-53"]
set_opts(skip_layers=None); net.reload_options()
base = [inflight() for _ in range(3)]
print("all: ", ["%.1f" % b for b in base], flush=True)
b0 = sorted(base)[1]
tot = 0
for g in groups:
    set_opts(skip_layers=g); net.reload_options()
    f = min(inflight(), inflight())
    set_opts(skip_layers=None); net.reload_options()
    print(f"skip {g:>6}: {f:7.1f} us/step  marginal {b0 - f:6.1f} us", flush=True)
base = [inflight() for _ in range(2)]
print("all again: ", ["%.1f" % b for b in base], flush=True)
