#!/usr/bin/env python3
"""ResNet-50 logits vs the oracle at batch sizes the test-suite does not use (1, 7, 33, 64): run on the GPU box."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tf2_amd import config as cfg, network, synth
from oracle import netref
t = cfg.resnet50_tables()
qv = np.loadtxt("tests/golden/resnet50_Q", dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
ref = netref.RefNet(t, qv, model)
for b, seed in ((1, 3), (7, 4), (33, 5), (64, 6)):
    x = synth.synth_images(t, b, seed)
    got = r.run_batch(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
    n = min(b, 9)
    want = ref.logits(ref.run(x[:n]))
    assert (got[:n] == want).all(), b
    if b > n:   # the rest: same image alone gives the same logits
        for i in (n, b - 1):
            alone = r.run_batch(torch.from_numpy(x[i:i + 1]).to("cuda:0")).cpu().numpy()
            assert (alone[0] == got[i]).all(), (b, i)
    print("batch", b, "ok")
