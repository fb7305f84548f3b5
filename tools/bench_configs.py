#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (synthetic weights / Q / images, one MI355X): images/s one batch at a
time and with four batches in flight.  bench.py measures the headline network; this is the breadth table of DESIGN.md section 5."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tf2_amd import config as cfg, network, synth

def measure(name, t, batch, seed, steps=20, inflight=4):
    q = synth.synth_q_values(t, seed, spread=1)
    model = synth.synth_model(t, q, seed)
    net = network.NetWork(t); net.Init(model, synth.q_text(q), device="cuda:0", pack_mode=0)
    x = torch.from_numpy(synth.synth_images(t, batch, seed)).to("cuda:0")
    r = network.Runner(None, net)
    t_spin = time.perf_counter() + 0.6           # out of the idle power state first (tools/clock_sample.py; bench.py --spinup-ms)
    while time.perf_counter() < t_spin:
        for _ in range(2): r.run_batch(x)
        torch.cuda.synchronize()
    for _ in range(3): r.run_batch(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): r.run_batch(x)
    torch.cuda.synchronize(); serial = batch * steps / (time.perf_counter() - t0)
    streams = [torch.cuda.Stream() for _ in range(inflight)]; runners = [network.Runner(None, net) for _ in range(inflight)]
    for i in range(2 * inflight):
        with torch.cuda.stream(streams[i % inflight]): runners[i % inflight].run_batch(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps * 2):
        with torch.cuda.stream(streams[k % inflight]): runners[k % inflight].run_batch(x)
    torch.cuda.synchronize(); fl = batch * steps * 2 / (time.perf_counter() - t0)
    print(f"{name:34s} batch {batch:3d}  {serial:9.0f} img/s one batch at a time   {fl:9.0f} img/s four in flight   packed image {len(net.packed_host())/1e6:6.1f} MB", flush=True)

measure("SqueezeNet 1.1, 227x227", cfg.squeezenet11_tables(), 32, 6)
measure("VGG16, 224x224", cfg.vgg16_tables(), 32, 1, steps=10)
measure("SSD300-VGG, 300x300, full width", cfg.ssd300_tables(), 32, 3, steps=10)
