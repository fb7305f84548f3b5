#!/usr/bin/env python3
"""Phase timeline of one conv_c3 launch: per block, 100 MHz wall-clock stamps (start, prologue landed, end of every chunk, K loop end,
end).  python tools/c3_timeline.py --net vgg16 --layer 8"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--net", default="vgg16"); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--layer", type=int, default=8)
a = ap.parse_args()
import torch
from tf2_amd import config as cfg, network, synth
t = {"vgg16": cfg.vgg16_tables, "ssd300": cfg.ssd300_tables}[a.net]()
qv = synth.synth_q_values(t, 0, spread=1)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x)
torch.cuda.synchronize()
row = [l for l in net.describe_launches(a.batch, 0) if l["layer"] == a.layer and "c3" in l["kernel"]][0]
print(row["kernel"])
nblk = row["grid"] * 8
dbg = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda:0")
set_opts(dbgptr2=str(dbg.data_ptr())); set_opts(dbglayer=str(a.layer))
net.reload_options()
for _ in range(2): r.run_batch(x)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16).astype(np.float64)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
print("blocks", len(d), "first start -> last end %.2f us; block lifetime median %.2f us" % ((d[:, 11].max() - t0) / 100, np.median(d[:, 11] - d[:, 0]) / 100))
prev = 0
for i, n in enumerate(["start", "prologue landed"] + ["chunk %d done" % c for c in range(8)] + ["K loop end", "end"]):
    col = d[:, i]
    if (col > 0).sum() == 0: continue
    m = np.median((col - d[:, 0])[col > 0]) / 100
    print(f"  {n:18s} {m:7.2f}   +{m - prev:.2f}")
    prev = m
