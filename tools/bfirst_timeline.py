#!/usr/bin/env python3
"""Phase timeline of the conv_bfirst launch (ResNet-50 rows 1-4): per block, 100 MHz wall-clock stamps of thread 0."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--conc", type=int, default=1)
a = ap.parse_args()
import torch
from tf2_amd import config as cfg, network, synth
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x, concurrency=a.conc)
torch.cuda.synchronize()
names = ["start", "prologue landed", "reduce -> halo", "3x3 loop end", "B tile written", "end"]
row = [l for l in net.describe_launches(a.batch, a.conc) if "conv_bfirst" in l["kernel"]]
if not row:
    print("no conv_bfirst launch in this plan"); sys.exit(0)
nblk = row[0]["grid"]
dbg = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda:0")
set_opts(dbgptr2=str(dbg.data_ptr())); set_opts(dbglayer=str(row[0]["layer"]))
net.reload_options()
for _ in range(2): r.run_batch(x, concurrency=a.conc)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16).astype(np.float64)
t0 = d[:, 0].min()
print(row[0]["kernel"], "blocks", nblk, "first start -> last end %.2f us; block starts spread over %.2f us; block life median %.2f us (max %.2f)" %
      ((d[:, 5].max() - t0) / 100, (d[:, 0].max() - t0) / 100, np.median(d[:, 5] - d[:, 0]) / 100, (d[:, 5] - d[:, 0]).max() / 100))
print("  " + " | ".join(f"{n} {np.median(d[:, i] - d[:, 0]) / 100:.2f}" for i, n in enumerate(names)))
late = d[:, 0] - t0 > 100 * 2.0
print("  blocks starting > 2 us after the first: %d; their life median %.2f us" % (late.sum(), np.median((d[late, 5] - d[late, 0])) / 100 if late.any() else 0))
