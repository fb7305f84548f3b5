#!/usr/bin/env python3
"""Per-block, per-tile cycle stamps of the persistent kernel (conv_mfma_p.hip) for one layer."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_amd import config as cfg, network, synth
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--layer", type=int, default=1)
a = ap.parse_args()
os.environ.setdefault("TF2_AMD_P", "1")
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x)
torch.cuda.synchronize()
dbg = torch.zeros(8192 * 16 + 64 * 16 * 8, dtype=torch.int64, device="cuda:0")
os.environ["TF2_AMD_DBGPTR2"] = str(dbg.data_ptr()); os.environ["TF2_AMD_DBGLAYER"] = str(a.layer)
r.run_batch(x); torch.cuda.synchronize()
allw = dbg.cpu().numpy()
d = allw[:8192 * 16].reshape(-1, 16)
w = allw[8192 * 16:].reshape(64, 16, 8)
d = d[d[:, 15] != 0]
print("blocks", len(d))
life = d[:, 15] - d[:, 0]
print("block lifetime ticks: min/median/max", life.min(), int(np.median(life)), life.max(), f"({np.median(life)/2400:.2f} us)")
# stamps of the block's SECOND tile: per step [entry, vmcnt wait done, barrier done, MFMAs issued], and in the
# last step additionally [phase shifts done, residual wait done, epilogue + stores issued]
st = d[:, 1:15].astype(np.int64)
ok = (st != 0).all(1) if False else (st[:, 0] != 0)
st = st[ok]; base = st[:, [0]]
rel = st - base; rel[st == 0] = -1
med = [int(np.median(rel[rel[:, i] >= 0, i])) if (rel[:, i] >= 0).any() else -1 for i in range(14)]
print("second tile, median stamps rel. to its first step entry:", med)
print("deltas:", [med[i + 1] - med[i] if med[i + 1] >= 0 else None for i in range(13)])

# per-wave view of the second tile's steps 0 and 1 (blocks 0..63): when each wave entered the wait, when its wait ended
for blk in (0, 1, 2):
    for stp in (0, 1):
        ww = w[blk, :, stp * 4:stp * 4 + 4]
        if ww[:, 0].min() == 0: continue
        t0 = ww[:, 0].min()
        print(f"block {blk} step {stp}: wave entry (rel) {[int(v - t0) for v in ww[:, 0]]}")
        print(f"                 wait done  (rel) {[int(v - t0) for v in ww[:, 1]]}  allowed-outstanding {ww[:, 2].tolist()}")
