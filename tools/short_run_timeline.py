#!/usr/bin/env python3
"""Where the time of a SHORT timed region goes (the driver times 20 steps after 5 warm-up steps): host clock at the end of every
step's enqueue and GPU timestamps (HIP events on the step's own stream) of every step's start and end, relative to the moment
the host starts enqueuing.  usage: short_run_timeline.py [steps] [inflight] [warmup]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from tf2_amd import config as cfg, network, synth, streams

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nfl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
use_feeder = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = "cuda:0"
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device=dev)
x = torch.from_numpy(synth.synth_images(t, 32, 100)).to(dev)
sts = [torch.cuda.Stream(device=dev) for _ in range(nfl)] if nfl > 1 else [torch.cuda.current_stream(dev)]
rns = [network.Runner(None, net) for _ in range(nfl)]
for st, rn in zip(sts, rns):
    with torch.cuda.stream(st):
        rn.run_batch(x)
torch.cuda.synchronize()

feeder = None
if use_feeder and nfl > 1:
    from tf2_amd.feeder import StreamFeeder
    feeder = StreamFeeder(sts, rns, dev)

def region(record):
    k0 = [0]
    def step():
        i = k0[0] % nfl; k0[0] += 1
        if feeder is not None:
            def work(rn, i=i):
                if record is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(sts[i])
                    h0 = time.perf_counter()
                rn.run_batch(x, concurrency=1)
                if record is not None:
                    e1.record(sts[i]); record.append((e0, e1, time.perf_counter(), h0, i))
            feeder.submit(i, work)
            return
        with torch.cuda.stream(sts[i]):
            if record is not None:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(sts[i])
                h0 = time.perf_counter()
            rns[i].run_batch(x, concurrency=1 if nfl > 1 else 0)
            if record is not None:
                e1.record(sts[i]); record.append((e0, e1, time.perf_counter(), h0, i))
    t_spin = time.perf_counter() + 0.6          # out of the idle power state first (tools/clock_sample.py)
    while time.perf_counter() < t_spin:
        for _ in range(4):
            step()
        if feeder is not None: feeder.drain()
        torch.cuda.synchronize()
    for _ in range(warm):
        step()
    if feeder is not None: feeder.drain()
    torch.cuda.synchronize()
    if record is not None:
        del record[:]
        base = torch.cuda.Event(enable_timing=True); base.record(torch.cuda.current_stream(dev))
        for st in sts:
            st.wait_event(base)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if feeder is not None: feeder.drain()
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t0, t_enq, t1, (base if record is not None else None)

for _ in range(2):
    t0, te, t1, _ = region(None)
    print(f"plain: {steps} steps in {(t1 - t0) * 1e6:.0f} us ({32 * steps / (t1 - t0):.0f} img/s), host enqueue done at {(te - t0) * 1e6:.0f} us")
rec = []
t0, te, t1, base = region(rec)
print(f"with events: {steps} steps in {(t1 - t0) * 1e6:.0f} us, host enqueue done at {(te - t0) * 1e6:.0f} us")
rec.sort(key=lambda r: r[2])
s = np.asarray([base.elapsed_time(r[0]) for r in rec]) * 1e3
e = np.asarray([base.elapsed_time(r[1]) for r in rec]) * 1e3
h = np.asarray([r[2] - t0 for r in rec]) * 1e6
h0 = np.asarray([r[3] - t0 for r in rec]) * 1e6
print(" stream host_enq_start host_enq_done  gpu_start   gpu_end  latency")
for k in range(len(rec)):
    print(f"{rec[k][4]:6d} {h0[k]:14.0f} {h[k]:13.0f} {s[k]:10.0f} {e[k]:9.0f} {e[k] - s[k]:8.0f}")
order = np.sort(e)
print("completion intervals:", " ".join(f"{v:.0f}" for v in np.diff(order)))
