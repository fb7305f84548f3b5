#!/usr/bin/env python3
"""Phase timeline of one conv_bgroup launch (option bgroup=1, the default one batch at a time): per block, 100 MHz wall-clock stamps at the phase boundaries."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
set_opts(bgroup="1"); set_opts(bgroup_min7="1"); set_opts(bgroup_min14="1"); set_opts(bgroup_min28="1"); set_opts(bgroup_min56f="1"); set_opts(bfirst="1");
import torch
from tf2_amd import config as cfg, network, synth
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--layer", type=int, default=31)
a = ap.parse_args()
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x, concurrency=0)
torch.cuda.synchronize()
dbg = torch.zeros(8 * a.batch * 16, dtype=torch.int64, device="cuda:0")
set_opts(dbgptr2=str(dbg.data_ptr())); set_opts(dbglayer=str(a.layer))
net.reload_options()
for _ in range(2): r.run_batch(x, concurrency=0)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)[:, :11].astype(np.float64)
t0 = d[:, 0].min()
names = ["start", "hdr+wA landed", "A loop end", "signal 1", "wait 1 done", "halo landed", "B loop end", "signal 2", "wait 2 done", "tile landed", "end"]
print("blocks", len(d), "first start -> last end %.2f us; block starts spread over %.2f us" % ((d[:, 10].max() - t0) / 100, (d[:, 0].max() - t0) / 100))
print("phase (median over blocks, us since the block's own start | median duration of the step):")
for i, n in enumerate(names):
    print(f"  {n:16s} {np.median(d[:, i] - d[:, 0]) / 100:7.2f}   {('+%.2f' % (np.median(d[:, i] - d[:, i - 1]) / 100)) if i else ''}")
