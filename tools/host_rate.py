#!/usr/bin/env python3
"""Host-side cost of enqueueing one step (56 launches through tf2_net_run) vs the GPU's step time: is the four-in-flight figure
bound by the launching thread?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tf2_amd import config as cfg, network, synth
t = cfg.resnet50_tables(); qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, 32, 1)).to("cuda:0")
streams = [torch.cuda.Stream(device="cuda:0") for _ in range(4)]
rs = [network.Runner(None, net) for _ in range(4)]
for k in range(12):
    with torch.cuda.stream(streams[k % 4]): rs[k % 4].run_batch(x, concurrency=1)
torch.cuda.synchronize()
for steps in (20, 100, 400):
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % 4]): rs[k % 4].run_batch(x, concurrency=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"steps {steps:>3}: host enqueue {(t1 - t0) / steps * 1e6:7.1f} us/step, total {(t2 - t0) / steps * 1e6:7.1f} us/step "
          f"({32 * steps / (t2 - t0):8.0f} img/s); the host was done {(t2 - t1) * 1e6:8.0f} us before the GPU", flush=True)
# the same with the queue kept short: is the enqueue cost different when the GPU is the one waiting?
r = rs[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(50):
    r.run_batch(x, concurrency=0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one stream: host enqueue {(t1 - t0) / 50 * 1e6:7.1f} us/step, total {(t2 - t0) / 50 * 1e6:7.1f} us/step")
