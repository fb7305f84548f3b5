#!/usr/bin/env python3
"""Phase timeline of one conv_bband launch: per block, 100 MHz wall-clock stamps at the phase boundaries."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--layer", type=int, default=31)
ap.add_argument("--rows", type=int, default=7); ap.add_argument("--conc", type=int, default=1)
a = ap.parse_args()
set_opts(bband="2"); set_opts(bband_rows=str(a.rows)); set_opts(bband_rows_alone=str(a.rows)); set_opts(bband_min="1")
import torch
from tf2_amd import config as cfg, network, synth
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x, concurrency=a.conc)
torch.cuda.synchronize()
nblk = [l for l in net.describe_launches(a.batch, a.conc) if l["layer"] == a.layer][0]["grid"]
dbg = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda:0")
set_opts(dbgptr2=str(dbg.data_ptr())); set_opts(dbglayer=str(a.layer))
net.reload_options()
for _ in range(2): r.run_batch(x, concurrency=a.conc)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)[:, :7].astype(np.float64)
t0 = d[:, 0].min()
names = ["start", "prologue landed", "reduce loop end", "halo tile done", "3x3 loop end", "B tile done", "expand end"]
print("blocks", len(d), "first start -> last end %.2f us; block starts spread over %.2f us" % ((d[:, 6].max() - t0) / 100, (d[:, 0].max() - t0) / 100))
for i, n in enumerate(names):
    print(f"  {n:18s} {np.median(d[:, i] - d[:, 0]) / 100:7.2f}   {('+%.2f' % (np.median(d[:, i] - d[:, i - 1]) / 100)) if i else ''}")
