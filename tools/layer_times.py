#!/usr/bin/env python3
"""Per-layer HIP-event timing table (ms, TOP/s, GB/s, grid) for ResNet50 at a given batch."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
import torch
from tf2_amd import config as cfg, network, synth, _lib
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_packed as emu

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--out", default=None)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--uniform-q", action="store_true", help="one Q value per tensor (every layer single-window): the bound for grouped packing")
ap.add_argument("--uniform-q2", action="store_true", help="the same, only for tensors that carry exactly two Q values")
a = ap.parse_args()
t = cfg.resnet50_tables(); plan = cfg.build_plan(t)
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
if a.uniform_q or a.uniform_q2:
    pos = 3
    for L in plan:
        if a.uniform_q or len(np.unique(qv[pos:pos + L.N])) == 2:
            qv[pos:pos + L.N] = int(np.round(qv[pos:pos + L.N].mean()))
        pos += L.N
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0", pack_mode=a.mode)
_, pls = emu.parse(net.packed_host())
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
dbg = torch.zeros(64 * 16, dtype=torch.int64, device="cuda:0")
if a.stamps:
    set_opts(dbgptr=str(dbg.data_ptr()))
    net.reload_options()
for _ in range(3): r.run_batch(x)
torch.cuda.synchronize()
_lib.check(_lib.lib().tf2_net_profile(net._h, 1))
for _ in range(a.steps): r.run_batch(x)
torch.cuda.synchronize()
n = len(plan); ms = np.zeros(n, np.float32); nl = np.zeros(n, np.int32); kd = np.zeros(n, np.int32)
_lib.check(_lib.lib().tf2_net_profile_read(net._h, ms.ctypes.data, nl.ctypes.data, kd.ctypes.data, n))
ms /= np.maximum(nl, 1)
rows = []
# table rows computed by another row's launch (conv_bneck pairs), from the library's own launch list
launches = net.describe_launches(a.batch, 0)
own = {l["layer"] for l in launches}
grid_of = {}
for l in launches:
    if "conv_" in l["kernel"]: grid_of.setdefault(l["layer"], l["grid"])
fused_into = {i: max(j for j in own if j < i) for i in range(n) if i not in own}
print(f"{'l':>2} {'k':>1} {'C':>4} {'N':>4} {'HW':>3} s P  ent/mt blocks   us     TOPS   GB/s")
for i, L in enumerate(plan):
    pl = pls[i]
    if i in fused_into:
        rows.append(dict(layer=i, k=L.k, C=L.C, N=L.N, HW=L.OH, stride=L.stride, fused_into=fused_into[i], us=0.0))
        print(f"{i:>2} {L.k} {L.C:>4} {L.N:>4} {L.OH:>3} {L.stride}  (computed by the launch of layer {fused_into[i]}: its time, ops and bytes are on that row)")
        continue
    fused = [j for j, f in fused_into.items() if f == i]
    ops = 2 * L.N * L.C * L.k * L.k * L.OH * L.OW * a.batch
    byts = (L.C * L.H * L.W + L.N * L.PH * L.PW * (2 if L.add_src >= 0 else 1)) * a.batch
    for j in fused:                      # a fused launch: both rows' ops; the intermediate tensor never reaches HBM
        F = plan[j]
        ops += 2 * F.N * F.C * F.k * F.k * F.OH * F.OW * a.batch
        byts += (F.N * F.PH * F.PW * (2 if F.add_src >= 0 else 1) - L.N * L.PH * L.PW) * a.batch
    TM = int(pl["TM"]); npix = a.batch * L.OH * L.OW
    blocks = grid_of.get(i, 0)          # the launch's actual grid
    ent = int(pl["n_entries"]) / max(1, int(pl["n_mtiles"]))
    us = max(ms[i] * 1e3, 1e-9)
    rows.append(dict(layer=i, k=L.k, C=L.C, N=L.N, HW=L.OH, stride=L.stride, phases=int(pl["n_phases"]), entries_per_mtile=ent,
                     blocks=blocks, us=float(us), tops=ops / us / 1e6, gbps=byts / us / 1e3))
    print(f"{i:>2} {L.k} {L.C:>4} {L.N:>4} {L.OH:>3} {L.stride} {int(pl['n_phases'])} {ent:6.1f} {blocks:>6} {us:7.1f} {ops/us/1e6:7.1f} {byts/us/1e3:7.1f}")
print("total us", float(ms.sum() * 1e3))
if a.stamps:
    d = dbg.cpu().numpy().reshape(64, 16)
    print("cycle stamps of block 0 (deltas, 100 MHz ticks?): start->e_start, ->header+A landed+barrier, ->prologue B issued, ->loop end, ->epilogue end")
    for i in range(n):
        r = d[i]
        print(i, [int(r[1] - r[0]), int(r[2] - r[1]), int(r[3] - r[2]), int(r[5] - r[3]), int(r[6] - r[5])],
              "K loop of block 0 / wave 0 (cycles per step): vmcnt wait %.0f  barrier %.0f  body(reads, DMA issue, MFMA issue) %.0f  over %d steps" %
              (r[8] / max(1, r[11]), r[9] / max(1, r[11]), r[10] / max(1, r[11]), r[11]) if r[11] else "")
if a.out:
    json.dump(rows, open(a.out, "w"), indent=0, default=float)
