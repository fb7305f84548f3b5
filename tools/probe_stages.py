#!/usr/bin/env python3
"""Where does a step's time go, one batch at a time and with four batches in flight?  Runs ResNet50 with all layers but one
stage left out (probe build: option skip_layers=lo-hi; results wrong, durations only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TF2_AMD_LIB"] = os.path.join(ROOT, "tf2_amd", "libtf2amd_probe.so"); os.environ["TF2_AMD_TOOL_LIB"] = "1"     # (built on demand: make -C tf2_amd/csrc probe)
from tf2_amd._lib import set_opts  # noqa: E402
import numpy as np, torch
from tf2_amd import config as cfg, network, synth, _lib
t = cfg.resnet50_tables(); plan = cfg.build_plan(t)
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, 32, 1)).to("cuda:0")
streams = [torch.cuda.Stream(device="cuda:0") for _ in range(4)]
def serial():
    r = network.Runner(None, net)
    for _ in range(10): r.run_batch(x, concurrency=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): r.run_batch(x, concurrency=0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e6
def inflight(n=4, steps=80):
    rs = [network.Runner(None, net) for _ in range(n)]
    for k in range(2 * n):
        with torch.cuda.stream(streams[k % n]): rs[k % n].run_batch(x, concurrency=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % n]): rs[k % n].run_batch(x, concurrency=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
ranges = {"all": None, "only prep+stem+pool (skip 1-53)": ["1-53"], "skip stem..stage2 (0-10)": ["0-10"], "skip stage3 (11-23)": ["11-23"],
          "skip stage4 (24-42)": ["24-42"], "skip stage5+fc (43-53)": ["43-53"], "skip 11-53 (stem+stage2 only)": ["11-53"],
          "skip 0-23 (stage 4,5 only)": ["0-23"], "skip 0-42 (stage 5 only)": ["0-42"]}
for name, rg in ranges.items():
    if rg: set_opts(skip_layers=rg[0])
    else: set_opts(skip_layers=None)
    s = serial(); f = inflight()
    print(f"{name:>36}: one at a time {s:7.1f} us/step   four in flight {f:7.1f} us/step", flush=True)
