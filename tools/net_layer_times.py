#!/usr/bin/env python3
"""Per-layer HIP-event times of any network (one batch at a time, launch plan of --conc) -- the library's own profile hook
(tf2_net_profile), the way bench.py reads it.  python tools/net_layer_times.py --net vgg16 [--batch 32] [--conc 0]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf2_amd import config as cfg, network, synth, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--net", default="vgg16"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--conc", type=int, default=0); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
t, q, seed, _, _ = synth.bench_network(a.net)          # the networks exactly as bench.py runs them
net = network.NetWork(t); net.Init(synth.synth_model(t, q, seed), synth.q_text(q), device="cuda:0", pack_mode=0)
dev = torch.device("cuda:0")
runner = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to(dev)
for _ in range(5): runner.run_batch(x, concurrency=a.conc)
_lib.check(_lib.lib().tf2_net_profile(net._h, 1))
for _ in range(a.steps): runner.run_batch(x, concurrency=a.conc)
torch.cuda.synchronize(dev)
n = len(net.plan)
ms = np.zeros(n, np.float32); nl = np.zeros(n, np.int32); kinds = np.zeros(n, np.int32)
_lib.check(_lib.lib().tf2_net_profile_read(net._h, ms.ctypes.data, nl.ctypes.data, kinds.ctypes.data, n))
_lib.check(_lib.lib().tf2_net_profile(net._h, 0))
names = {}
for r in net.describe_launches(a.batch, a.conc):
    names.setdefault(r["layer"], []).append(r["kernel"][:70] + " grid %d" % r["grid"])
tot = 0.0
for i, L in enumerate(net.plan):
    if nl[i] == 0: continue
    us = 1e3 * ms[i] / nl[i] * (nl[i] / a.steps); tot += us
    gop = 2.0 * a.batch * L.OH * L.OW * L.N * L.C * L.k * L.k / 1e9
    print("%2d k%d %4d->%4d %3dx%-3d %7.1f us %6.0f TOP/s  %s" % (i, L.k, L.C, L.N, L.OH, L.OW, us, gop / us * 1e3 / 1e3 * 1e0, " | ".join(names.get(i, []))))
print("sum %.1f us" % tot)
