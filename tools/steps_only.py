#!/usr/bin/env python3
"""N steps of ResNet50 (or --net: another BASELINE.json network) at one batch size, one batch at a time on one stream, and nothing else (no latency leg, no sweep, no
event records): the workload rocprofv3 --kernel-trace --stats and the --pmc passes are run on (tools/round_evidence.sh), so
that every row of their per-kernel tables is this configuration only."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tf2_amd import config as cfg, network, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=0, help="untimed steps are kernels too: keep 0 for profiles, the summary divides by --steps")
ap.add_argument("--conc", type=int, default=0, help="launch plan: 0 one batch at a time, 1 the several-batches-in-flight choice")
ap.add_argument("--spinup-ms", type=float, default=600.0,
                help="keep the device busy this long with torch matrix products first (other kernel names: they do not enter the tf2 rows "
                     "of the profile): an idle MI355X needs ~0.4 s of load to reach its engine clock (tools/clock_sample.py)")
ap.add_argument("--net", default="resnet50", choices=["resnet50", "squeezenet", "vgg16", "ssd300", "googlenet", "resnet50_pruned"])
ap.add_argument("--meta", default=None, help="write {batch, steps, launches} here")
a = ap.parse_args()
t, qv, seed, net_name, _ = synth.bench_network(a.net)
net = network.NetWork(t); net.Init(synth.synth_model(t, qv, seed), synth.q_text(qv), device="cuda:0")
r = network.Runner(None, net)
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
if a.spinup_ms > 0:
    import time
    m = torch.randn(4096, 4096, device="cuda:0", dtype=torch.float16)
    t_end = time.perf_counter() + a.spinup_ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(20):
            m2 = m @ m
        torch.cuda.synchronize()
    del m, m2
for _ in range(a.warmup + a.steps):
    r.run_batch(x, concurrency=a.conc)
torch.cuda.synchronize()
if a.meta:
    conv_rows = [l for l, L in enumerate(cfg.build_plan(t)) if not L.ipool]        # table rows that are convolutions (their pool / average launches included)
    json.dump(dict(net=net_name, batch=a.batch, steps=a.warmup + a.steps, conc=a.conc, conv_rows=conv_rows, launches=net.describe_launches(a.batch, a.conc)), open(a.meta, "w"), indent=0)
