#!/usr/bin/env python3
"""Per-block start/end cycle stamps + CU ids of one conv layer (conv_mfma2 kernel): concurrency and lifetimes."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
import torch
from tf2_amd import config as cfg, network, synth
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--layer", type=int, default=1)
a = ap.parse_args()
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
for _ in range(3): r.run_batch(x)
torch.cuda.synchronize()
dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda:0")
set_opts(dbgptr2=str(dbg.data_ptr())); set_opts(dbglayer=str(a.layer))
net.reload_options()
r.run_batch(x); torch.cuda.synchronize()
from tf2_amd import _lib
_lib.check(_lib.lib().tf2_net_profile(net._h, 1))
r.run_batch(x); torch.cuda.synchronize()
nl_ = len(cfg.build_plan(t)); ms = np.zeros(nl_, np.float32); nn = np.zeros(nl_, np.int32); kd = np.zeros(nl_, np.int32)
_lib.check(_lib.lib().tf2_net_profile_read(net._h, ms.ctypes.data, nn.ctypes.data, kd.ctypes.data, nl_))
print(f"layer {a.layer}: event-timed kernel {ms[a.layer] / max(nn[a.layer], 1) * 1e3:.2f} us")
d = dbg.cpu().numpy().reshape(-1, 8)
d = d[d[:, 1] != 0]
n = len(d)
t0 = d[:, 0].min()
st, en = d[:, 0] - t0, d[:, 1] - t0
hw = d[:, 2]; xcc = np.zeros(len(d), np.int64)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
print("blocks", n, "kernel span cycles", en.max(), "distinct CUs", len(set(cuid.tolist())))
life = en - st
seg = np.stack([d[:,4]-d[:,0], d[:,5]-d[:,4], d[:,6]-d[:,5], d[:,1]-d[:,6]], 1)
w0, w1 = d[:, 3], d[:, 7]
print(f"RAW wall ticks: first start {w0.min()} last end {w1.max()}")
print(f'wall clock (100 MHz): first block start -> last block end {(w1.max() - w0.min()) / 100:.2f} us; first->last block START {(w0.max() - w0.min()) / 100:.2f} us; median block {np.median(w1 - w0) / 100:.2f} us')
print("segment medians (start->gather words, ->prologue DMAs issued, ->hdr+stage0 landed, ->end):", np.median(seg,0).astype(int).tolist())
print("segment means:", seg.mean(0).astype(int).tolist())
first = np.argsort(d[:,0])[:256]
print("segments, mean of 256 earliest blocks:", seg[first].mean(0).astype(int).tolist())
print("lifetime cycles: min/median/mean/max", life.min(), int(np.median(life)), int(life.mean()), life.max())
# per-CU analysis (the cycle counters are per CU: only stamps of one CU are comparable)
spans = []; concs = []; example = None
for cu_ in sorted(set(cuid.tolist())):
    m = cuid == cu_
    s0 = d[m, 0].min(); sx = d[m, 0] - s0; ex = d[m, 1] - s0
    if ex.max() > 10_000_000: continue          # counters differing inside a "CU" id: skip
    ev = sorted([(t_, 1) for t_ in sx] + [(t_, -1) for t_ in ex])
    cur = 0; area = 0; last = 0
    for tm, dlt in ev:
        area += cur * (tm - last); last = tm; cur += dlt
    spans.append(ex.max()); concs.append(area / ex.max())
    if example is None: example = sorted(zip(sx.tolist(), ex.tolist()))
spans = np.array(spans)
if len(spans) == 0: sys.exit(0)
print(f"per-CU: {len(spans)} CUs usable; span ticks min/median/max {spans.min()} {int(np.median(spans))} {spans.max()} ({np.median(spans)/2400:.2f} us median); mean concurrent blocks per CU {np.mean(concs):.2f}")
print("one CU's blocks (start, end):", example)
