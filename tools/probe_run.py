#!/usr/bin/env python3
"""Which pipe binds a conv layer?  Per-layer HIP-event times (one batch at a time) and the four-batches-in-flight rate of
ResNet50 with ONE component of the ring kernels left out at a time: activation DMAs, weight DMAs, MFMAs, the epilogue,
the bounds checks of padded taps, the block barrier.  Needs the probe build (make -C tf2_amd/csrc probe ->
tf2_amd/libtf2amd_probe.so, the same sources with -DTF2_PROBES); the results of a probed run are WRONG by construction,
only durations are read.  The product library has no probe code (tf2_device.h TF2_PROBE_WORD)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["TF2_AMD_LIB"] = os.path.join(ROOT, "tf2_amd", "libtf2amd_probe.so"); os.environ["TF2_AMD_TOOL_LIB"] = "1"     # (built on demand: make -C tf2_amd/csrc probe)
import numpy as np
import torch
from tf2_amd import config as cfg, network, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--inflight-steps", type=int, default=60)
ap.add_argument("--out", default=None)
ap.add_argument("--last-layer", type=int, default=53, help="print per-layer rows up to this layer")
a = ap.parse_args()

PROBES = [("base", 0), ("noB", 8), ("noA", 16), ("noAB", 24), ("noMFMA", 32), ("noEpi", 64), ("noBar", 256),
          ("noAB_noMFMA", 56), ("noMFMA_noEpi", 96), ("prologue_only", 120), ("noStore", 2048), ("noEpi_noStore", 2048 + 64), ("exit_at_entry", 512), ("exit_after_hdr_words", 1024)]
t = cfg.resnet50_tables(); plan = cfg.build_plan(t)
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
n = len(plan)
streams = [torch.cuda.Stream(device="cuda:0") for _ in range(4)]


def layer_table(conc):
    set_opts(alt_conc="1" if conc else "0")
    net.reload_options()
    r = network.Runner(None, net)
    for _ in range(3): r.run_batch(x)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().tf2_net_profile(net._h, 1))
    for _ in range(a.steps): r.run_batch(x)
    torch.cuda.synchronize()
    ms = np.zeros(n, np.float32); nl = np.zeros(n, np.int32); kd = np.zeros(n, np.int32)
    _lib.check(_lib.lib().tf2_net_profile_read(net._h, ms.ctypes.data, nl.ctypes.data, kd.ctypes.data, n))
    _lib.check(_lib.lib().tf2_net_profile(net._h, 0))
    return ms / np.maximum(nl, 1) * 1e3


def serial_wall():
    set_opts(alt_conc="0")
    net.reload_options()
    r = network.Runner(None, net)
    for _ in range(3): r.run_batch(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): r.run_batch(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e6


def inflight_rate():
    set_opts(alt_conc="1")
    net.reload_options()
    rs = [network.Runner(None, net) for _ in streams]
    for st, r in zip(streams, rs):
        with torch.cuda.stream(st): r.run_batch(x)
    torch.cuda.synchronize()
    for k in range(8):
        with torch.cuda.stream(streams[k % 4]): rs[k % 4].run_batch(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.inflight_steps):
        with torch.cuda.stream(streams[k % 4]): rs[k % 4].run_batch(x)
    torch.cuda.synchronize()
    return a.batch * a.inflight_steps / (time.perf_counter() - t0)


res = {}
for name, bits in PROBES:
    set_opts(exp=str(bits))
    ser = layer_table(False)
    con = layer_table(True)
    rate = inflight_rate()
    sw = serial_wall()
    res[name] = dict(serial_wall_us=sw, serial_us=[float(v) for v in ser], conc_plan_us=[float(v) for v in con], inflight_img_s=float(rate))
    print(f"{name:>14}: serial sum {ser.sum():7.1f} us   conc-plan sum {con.sum():7.1f} us   4 in flight {rate:9.0f} img/s   serial wall {sw:7.1f} us/step", flush=True)

print("\nper layer, one batch at a time (us): " + " ".join(f"{p[0]:>9}" for p in PROBES))
for i, L in enumerate(plan):
    if i > a.last_layer: break
    print(f"{i:>2} k{L.k} C{L.C:<4} N{L.N:<4} {L.OH:>3}^2 s{L.stride} " + " ".join(f"{res[p[0]]['serial_us'][i]:9.1f}" for p in PROBES))
print("\nper layer, the several-streams launch plan on one stream (us): " + " ".join(f"{p[0]:>9}" for p in PROBES))
for i, L in enumerate(plan):
    if i > a.last_layer: break
    print(f"{i:>2} k{L.k} C{L.C:<4} N{L.N:<4} {L.OH:>3}^2 s{L.stride} " + " ".join(f"{res[p[0]]['conc_plan_us'][i]:9.1f}" for p in PROBES))
if a.out:
    json.dump(res, open(a.out, "w"))
