#!/usr/bin/env python3
"""Stress of the group launches (option bgroup=1, the default one batch at a time): many back-to-back steps at several batch sizes, logits against the plain
launch sequence every time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
set_opts(bgroup_min7="1"); set_opts(bgroup_min14="1"); set_opts(bgroup_min28="1"); set_opts(bgroup_min56f="1"); set_opts(bfirst="1");      # every batch size takes the group launches here
import torch
from tf2_amd import config as cfg, network, synth
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for B in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (32, 1, 7, 64, 9, 32)):
    x = torch.from_numpy(synth.synth_images(t, B, B)).to("cuda:0")
    set_opts(bgroup="0"); net.reload_options()
    ref = network.Runner(None, net).run_batch(x, concurrency=0).clone(); torch.cuda.synchronize()
    set_opts(bgroup="1"); net.reload_options()
    r = network.Runner(None, net)
    assert any("bgroup" in l["kernel"] for l in net.describe_launches(B, 0))
    bad = 0; t0 = time.perf_counter()
    for i in range(reps):
        y = r.run_batch(x, concurrency=0)
        if i % 10 == 9 or i < 3:
            torch.cuda.synchronize()
            bad += int(not torch.equal(y, ref))
    torch.cuda.synchronize()
    print(f"batch {B}: {reps} steps, {bad} mismatches, {B * reps / (time.perf_counter() - t0):.0f} img/s", flush=True)
