#!/usr/bin/env python3
"""Which shared resource binds the in-flight step?  The probe build (-DTF2_PROBES: results WRONG, durations only) with one component left
out of every kernel that implements the probe bit (ring kernels, conv_pw, conv_bneck, conv_stem): stores, epilogue arithmetic, MFMAs, loop DMAs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TF2_AMD_LIB", os.path.join(ROOT, "tf2_amd", "libtf2amd_probe.so")); os.environ["TF2_AMD_TOOL_LIB"] = "1"
from tf2_amd._lib import set_opts  # noqa: E402
import numpy as np, torch
from tf2_amd import config as cfg, network, synth
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, 32, 1)).to("cuda:0")
streams = [torch.cuda.Stream(device="cuda:0") for _ in range(4)]
def inflight(n=4, steps=160):
    rs = [network.Runner(None, net) for _ in range(n)]
    for k in range(4 * n):
        with torch.cuda.stream(streams[k % n]): rs[k % n].run_batch(x, concurrency=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % n]): rs[k % n].run_batch(x, concurrency=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
for _ in range(3): inflight()
PROBES = [("base", 0), ("noStore", 2048), ("noEpi", 64), ("noEpi_noStore", 2048 + 64), ("noMFMA", 32), ("noB (activation DMAs)", 8), ("noA (weight DMAs)", 16), ("noAB", 24), ("base", 0)]
for rep in range(2):
    for name, bits in PROBES:
        set_opts(exp=str(bits) if bits else None); net.reload_options()
        v = sorted(inflight() for _ in range(3))
        print(f"{name:28s} {v[0]:7.1f} {v[1]:7.1f} {v[2]:7.1f} us/step", flush=True)
