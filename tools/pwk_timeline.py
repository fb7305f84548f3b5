#!/usr/bin/env python3
"""Per-block wall-clock stamps (100 MHz) of one conv_pwk launch: where a block's life goes.
stamps: 0 entry | per tile it < 3: 1 + 3 it at the tile's wait, 2 + 3 it behind vmcnt(0), 3 + 3 it behind the barrier | 10 behind the last store's issue | 11 stores drained"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
import torch
from tf2_amd import config as cfg, network, synth
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--layers", default="5,11,12,14,27"); ap.add_argument("--opts", default="")
a = ap.parse_args()
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
os.environ["TF2_AMD_TEST"] = "1"
set_opts(pwk="2", pwk_minpix="0")
for kv in a.opts.split(","):
    if kv: set_opts(**{kv.split("=")[0]: kv.split("=")[1]})
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x)
torch.cuda.synchronize()
for layer in [int(v) for v in a.layers.split(",")]:
    dbg = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda:0")
    set_opts(dbgptr2=str(dbg.data_ptr())); set_opts(dbglayer=str(layer))
    net.reload_options()
    for _ in range(2):
        dbg.zero_(); r.run_batch(x); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 16)
    d = d[d[:, 0] != 0]
    if not len(d): print("layer", layer, "no stamps"); continue
    t0 = d[:, 0].min()
    us = lambda v: float(np.median(v)) / 100.0
    n_t = 1 + int((d[:, 4] != 0).any()) + int((d[:, 7] != 0).any())
    line = f"layer {layer}: {len(d)} blocks, first entry -> last drained {(d[:, 11].max() - t0) / 100:.2f} us | entry spread {(d[:, 0].max() - t0) / 100:.2f} us | block life median {us(d[:, 11] - d[:, 0]):.2f} us:"
    line += f" entry -> tile 0 wait {us(d[:, 1] - d[:, 0]):.2f} | wait {us(d[:, 2] - d[:, 1]):.2f} | barrier {us(d[:, 3] - d[:, 2]):.2f}"
    prev = 3
    for it in range(1, 3):
        m = d[:, 1 + 3 * it] != 0
        if not m.any(): break
        line += f" | tile {it - 1} compute {us(d[m, 1 + 3 * it] - d[m, prev]):.2f} | wait {us(d[m, 2 + 3 * it] - d[m, 1 + 3 * it]):.2f} | barrier {us(d[m, 3 + 3 * it] - d[m, 2 + 3 * it]):.2f}"
        prev = 3 + 3 * it
    last = np.where(d[:, 9] != 0, d[:, 9], np.where(d[:, 6] != 0, d[:, 6], d[:, 3]))
    line += f" | last tile compute {us(d[:, 10] - last):.2f} | store drain {us(d[:, 11] - d[:, 10]):.2f}"
    print(line, flush=True)
