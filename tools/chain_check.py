#!/usr/bin/env python3
"""Chain launches (TF2_AMD_CHAIN=1, conv_mfma2_chain_kernel) against the plain launch sequence on the same images: logits of
repeated runs must be identical, one batch at a time and with the several-streams plan; then rates of both."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from tf2_amd import config as cfg, network, synth, streams as tstreams

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
dev = torch.device("cuda:0")
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0", pack_mode=0)
x = torch.from_numpy(synth.synth_images(t, a.batch, 3)).to(dev)

def set_chain(v):
    os.environ["TF2_AMD_CHAIN"] = str(v)
    net.reload_options()

def rate(conc, n_streams, partition):
    ss = tstreams.partitioned_streams(n_streams, dev) if (partition and n_streams > 1) else [torch.cuda.Stream(dev) for _ in range(n_streams)]
    rs = [network.Runner(None, net) for _ in range(n_streams)]
    for w in range(2 * n_streams):
        with torch.cuda.stream(ss[w % n_streams]): rs[w % n_streams].run_batch(x, concurrency=conc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        with torch.cuda.stream(ss[i % n_streams]): rs[i % n_streams].run_batch(x, concurrency=conc)
    torch.cuda.synchronize()
    return a.batch * a.steps / (time.perf_counter() - t0)

for conc in (0, 1):
    set_chain(0)
    r = network.Runner(None, net)
    ref = r.run_batch(x, concurrency=conc).clone(); torch.cuda.synchronize()
    set_chain(1)
    n_launch = len(net.describe_launches(a.batch, conc))
    r2 = network.Runner(None, net)
    bad = 0
    for i in range(a.reps):
        y = r2.run_batch(x, concurrency=conc); torch.cuda.synchronize()
        if not torch.equal(y, ref): bad += 1
    print(f"concurrency {conc}: {n_launch} launches per step with chains; {bad} of {a.reps} runs differ from the plain sequence", flush=True)
for chain in (0, 1, 0, 1):
    set_chain(chain)
    print(f"chain {chain}: one batch at a time {rate(0, 1, False):9.0f} img/s (plan 0) {rate(1, 1, False):9.0f} (plan 1 alone) | "
          f"four in flight, XCD partitions {rate(1, 4, True):9.0f} | four plain streams {rate(1, 4, False) if not chain else float('nan'):9.0f}", flush=True)
