#!/usr/bin/env python3
"""One batch at a time, host-synchronised: microseconds per batch of ResNet50 at small batch sizes (tile-selection thresholds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from tf2_amd import config as cfg, network, synth
t = cfg.resnet50_tables(); q = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, q, 0), synth.q_text(q), device="cuda:0", pack_mode=0)
out = []
for b in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16").split(",")]:
    x = torch.from_numpy(synth.synth_images(t, b, 1)).to("cuda:0"); r = network.Runner(None, net)
    for _ in range(10): r.run_batch(x)
    torch.cuda.synchronize(); n = 60; t0 = time.perf_counter()
    for _ in range(n): r.run_batch(x); torch.cuda.synchronize()
    out.append("b%d %.1f" % (b, (time.perf_counter() - t0) / n * 1e6))
print(os.environ.get("TF2_AMD_OPTS", "default"), " ".join(out))
