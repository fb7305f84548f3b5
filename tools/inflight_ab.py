#!/usr/bin/env python3
"""A/B of launch-plan switches in ONE process: ResNet50, batch N, several batches in flight on plain streams (as
bench.py times them) and one batch at a time, per setting of the library's option string TF2_AMD_OPTS (re-read through tf2_net_reload_options).
Usage: inflight_ab.py --set "bband=0" --set "bband=1,bband_rows=7" ...
Prints img/s and whether the logits equal the first setting's."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tf2_amd import config as cfg, network, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--inflight", type=int, default=4)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--serial", type=int, default=1, help="also time one batch at a time")
ap.add_argument("--prio", type=str, default="", help="comma-separated stream priorities (torch: -1 high, 0 normal), e.g. -1,0,0,0")
a = ap.parse_args()
settings = a.set or ["bband=0", "bband=1"]
t = cfg.resnet50_tables()
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, 0), synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
prio = [int(v) for v in a.prio.split(",")] if a.prio else [0] * a.inflight
sts = [torch.cuda.Stream(device="cuda:0", priority=prio[i % len(prio)]) for i in range(a.inflight)]
# device spin-up (an idle MI355X needs ~0.4 s of load to reach its clock)
m = torch.randn(4096, 4096, device="cuda:0", dtype=torch.float16)
t_end = time.perf_counter() + 0.6
while time.perf_counter() < t_end:
    for _ in range(20): m2 = m @ m
    torch.cuda.synchronize()
ref = None
touched = set()
for rep in range(a.reps):
    for s in settings:
        # one setting = one TF2_AMD_OPTS string (csrc/opts.h), e.g. "bband=0" or "bband=1,bband_rows=7"
        os.environ["TF2_AMD_TEST"] = "1"; os.environ["TF2_AMD_OPTS"] = s.lower()
        net.reload_options()
        rs = [network.Runner(None, net) for _ in sts]
        def loop(n, conc):
            for i in range(n):
                with torch.cuda.stream(sts[i % len(sts)]): rs[i % len(sts)].run_batch(x, concurrency=conc)
        loop(2 * len(sts), 1); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(a.steps, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        got = [r._logits.clone() for r in rs]
        if ref is None: ref = got[0].clone()
        ok = all(bool((g == ref).all()) for g in got)
        line = f"{s:40s} in flight x{len(sts)}: {a.batch * a.steps / dt:9.0f} img/s  launches {len(net.describe_launches(a.batch, 1)):3d}  same logits {ok}"
        if a.serial:
            r0 = network.Runner(None, net)
            for _ in range(5): r0.run_batch(x, concurrency=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps): r0.run_batch(x, concurrency=0)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            line += f" | one at a time: {a.batch * a.steps / dt:9.0f} img/s  launches {len(net.describe_launches(a.batch, 0)):3d}  same logits {bool((r0._logits == ref).all())}"
        print(line, flush=True)
