#!/usr/bin/env python3
"""Per-launch HIP-event times of one step on ONE stream restricted to 8 / N XCDs (tf2_amd.streams): what one of N in-flight
batches sees on its own partition, without the other batches' interference.  rocprofv3 ignores the CU mask (its kernel durations on a
masked stream equal the unmasked ones), hence events (each pair adds ~1.7 us of record handling)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_amd import config as cfg, network, synth, streams, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--partition", type=int, default=4); ap.add_argument("--conc", type=int, default=1)
a = ap.parse_args()
t = cfg.resnet50_tables(); plan = cfg.build_plan(t)
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
st = streams.partitioned_streams(a.partition, "cuda:0")[0] if a.partition > 0 else torch.cuda.Stream(device="cuda:0")
import time
with torch.cuda.stream(st):
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for _ in range(10): r.run_batch(x, concurrency=a.conc)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps): r.run_batch(x, concurrency=a.conc)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e6
    _lib.check(_lib.lib().tf2_net_profile(net._h, 1))
    for _ in range(a.steps): r.run_batch(x, concurrency=a.conc)
    torch.cuda.synchronize()
n = len(plan); ms = np.zeros(n, np.float32); nl = np.zeros(n, np.int32); kd = np.zeros(n, np.int32)
_lib.check(_lib.lib().tf2_net_profile_read(net._h, ms.ctypes.data, nl.ctypes.data, kd.ctypes.data, n))
_lib.check(_lib.lib().tf2_net_profile(net._h, 0))
ms /= np.maximum(nl, 1)
tot = 0.0
for l in net.describe_launches(a.batch, a.conc):
    if l["layer"] < 0: continue
    first = not any(m["layer"] == l["layer"] and m is not l for m in net.describe_launches(a.batch, a.conc)[:0])
    print(f"row {l['layer']:3d} {ms[l['layer']] * 1e3:8.1f} us  grid {l['grid']:5d}  {l['kernel']}")
    tot += ms[l["layer"]] * 1e3
print(f"partition 1/{a.partition} conc-plan {a.conc}: wall per step {wall:.1f} us; sum of event pairs {tot:.1f} us")
