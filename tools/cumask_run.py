#!/usr/bin/env python3
"""Batches in flight on DISJOINT CU partitions: every in-flight stream is created with hipExtStreamCreateWithCUMask, so the
kernels of one batch keep to their own quarter (or half, eighth) of the chip instead of competing with the other batches'
kernels for the same CUs.  Prints img/s for ResNet50 at batch 32 per partitioning scheme and step count."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf2_amd._lib import set_opts  # noqa: E402
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from tf2_amd import config as cfg, network, synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=str, default="20,100")
ap.add_argument("--out", default=None)
a = ap.parse_args()

hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = C.CDLL(name); break
    except OSError:
        pass
assert hip is not None
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (C.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), words, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device="cuda:0")


def scheme(name, k, nparts):
    if name == "contig":                 # CUs [k*NCU/n, (k+1)*NCU/n)
        return [i for i in range(NCU) if i * nparts // NCU == k]
    if name == "mod":                    # CU i -> partition i % n
        return [i for i in range(NCU) if i % nparts == k]
    if name == "xcdrr":                  # if mask bit i is XCD i % 8: partition = a set of whole XCDs
        per = 8 // nparts
        return [i for i in range(NCU) if (i % 8) // per == k]
    raise ValueError(name)


t = cfg.resnet50_tables(); plan = cfg.build_plan(t)
qv = np.loadtxt(os.path.join(ROOT, "tests/golden/resnet50_Q"), dtype=np.int32)
model = synth.synth_model(t, qv, 0)
net = network.NetWork(t); net.Init(model, synth.q_text(qv), device="cuda:0")
x = torch.from_numpy(synth.synth_images(t, a.batch, 1)).to("cuda:0")
ref = network.Runner(None, net).run_batch(x).clone()
torch.cuda.synchronize()


def rate(streams, steps, conc):
    set_opts(alt_conc=str(conc))
    net.reload_options()
    k = len(streams)
    rs = [network.Runner(None, net) for _ in streams]
    for st, r in zip(streams, rs):
        with torch.cuda.stream(st): r.run_batch(x)
    torch.cuda.synchronize()
    for i in range(2 * k):
        with torch.cuda.stream(streams[i % k]): rs[i % k].run_batch(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % k]): rs[i % k].run_batch(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(bool((r._logits == ref).all()) for r in rs)
    return a.batch * steps / dt, ok


out = {}
steps_list = [int(s) for s in a.steps.split(",")]
plain4 = [torch.cuda.Stream(device="cuda:0") for _ in range(4)]
for steps in steps_list:
    r, ok = rate(plain4, steps, 1)
    out[f"plain_4_steps{steps}"] = r
    print(f"plain streams x4           steps {steps:>3}: {r:9.0f} img/s  parity {ok}", flush=True)
for sch, nparts, per in (("xcdrr", 4, 1), ("mod", 4, 1), ("xcdrr", 4, 2), ("xcdrr", 2, 2), ("xcdrr", 2, 3), ("xcdrr", 2, 4), ("xcdrr", 1, 4)):
    sts = []
    for rep in range(per):
        sts += [masked_stream(scheme(sch, k, nparts)) for k in range(nparts)]
    for steps in steps_list:
        for conc in (0, 1):
            r, ok = rate(sts, steps, conc)
            out[f"{sch}_{nparts}x{per}_steps{steps}_conc{conc}"] = r
            print(f"{sch:>6} {nparts} partitions x {per} streams  steps {steps:>3} conc-plan {conc}: {r:9.0f} img/s  parity {ok}", flush=True)
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
