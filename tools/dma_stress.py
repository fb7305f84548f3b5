#!/usr/bin/env python3
"""Stress of the LDS-DMA kernels under load (round 5).

conv_bband / conv_c3 / conv_c3_w9 issue their LDS-DMAs as inline assembly and tell a chunk's DMAs from younger loads by counted
`s_waitcnt vmcnt(N)` waits (csrc/vm_track.h); a wait that is one step too generous is bit-exact on an idle chip and wrong when the
DMAs land later -- i.e. under load.  This tool runs every instantiation of those kernels (and of conv_fc, whose partial sums go
through a scratch area) that `tf2_net_describe_launches` emits for the BASELINE.json networks at batches 1..64, in both launch plans,
BESIDE a hog on a second stream (a 512 MB device copy = HBM, and a 2 MB read-modify-write loop = L2), `--iters` steps each, and
compares the logits of EVERY step with a serial run of the same input on an otherwise idle chip.

  python tools/dma_stress.py [--iters 200] [--nets resnet50,vgg16,ssd300,squeezenet] [--out gpurun_out/dma_stress.txt]

With TF2_AMD_LIB pointing at the -DTF2_CHECK_DMA build (make -C tf2_amd/csrc check) the kernels also stamp a sentinel into every
chunk buffer before its DMA and count consumer reads that still see it (`tf2_check_dma_errors`); the tool prints that count."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

DMA_KERNELS = ("conv_bband_kernel", "conv_bfirst_kernel", "conv_bneck_kernel", "conv_c3_kernel", "conv_c3_w9_kernel", "fc_partial_kernel", "fc4_partial_kernel", "conv_fire_kernel", "conv_first_kernel",
               "conv_pwk_kernel", "conv_pwk_pair_kernel")      # (round 6: the persistent pointwise kernel's hand-written DMA waits)
KSP_MARK = "K over"                                           # ... and the split-K launches that exchange partial tiles between blocks (conv_mfma_sk KSP)
BATCHES = (1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 31, 32, 33, 48, 63, 64)


def networks(names):
    from tf2_amd import synth
    return [(n,) + synth.bench_network(n)[:3] for n in names]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--nets", default="resnet50,vgg16,ssd300,squeezenet")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dma_stress.txt"))
    ap.add_argument("--opts", default="", help="extra passes with other option strings, ';'-separated (e.g. 'bband_rows=4;bband=2')")
    args = ap.parse_args()
    import torch
    from tf2_amd import _lib, network, synth
    dev = torch.device("cuda", 0)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say(f"# dma_stress: library {_lib.LIB_PATH}, {args.iters} steps per case, hog = 512 MB copy + 2 MB read-modify-write on a second stream")
    hog_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev).random_(0, 255)
    hog_b = torch.empty_like(hog_a)
    hog_s = torch.zeros(2 << 20, dtype=torch.uint8, device=dev)
    hog_stream, run_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    total_bad = total_cases = 0
    seen_all = set()
    for opt_pass in [""] + [o for o in args.opts.split(";") if o]:
        if opt_pass:
            os.environ["TF2_AMD_OPTS"] = opt_pass
            os.environ["TF2_AMD_TEST"] = "1"
        for name, t, q, seed in networks([n for n in args.nets.split(",") if n] if not opt_pass else ["resnet50"]):
            net = network.NetWork(t)
            net.Init(synth.synth_model(t, q, seed), synth.q_text(q), device="cuda:0")
            seen = set()
            for B in BATCHES:
                for conc in (1, 0):
                    rows = [r for r in net.describe_launches(B, conc) if r["kernel"].startswith(DMA_KERNELS) or KSP_MARK in r["kernel"]]
                    keys = {(r["kernel"].split(" (")[0], r["grid"]) for r in rows}
                    if not keys - seen:
                        continue
                    new = sorted(keys - seen)
                    seen |= keys
                    x = torch.from_numpy(synth.synth_images(t, B, 900 + B)).to(dev)
                    ser = network.Runner(None, net)
                    ref = ser.run_batch(x, concurrency=conc).clone()
                    torch.cuda.synchronize()
                    rn = network.Runner(None, net)
                    outs = []
                    t0 = time.perf_counter()
                    with torch.cuda.stream(run_stream):
                        rn.run_batch(x, concurrency=conc)
                    torch.cuda.synchronize()
                    for i in range(args.iters):
                        with torch.cuda.stream(hog_stream):
                            hog_b.copy_(hog_a, non_blocking=True)
                            for _ in range(4):
                                hog_s.add_(1)
                        with torch.cuda.stream(run_stream):
                            outs.append(rn.run_batch(x, concurrency=conc).clone())
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    bad = sum(int(not torch.equal(o, ref)) for o in outs)
                    total_bad += bad
                    total_cases += 1
                    say(f"{name:10s} {('[' + opt_pass + '] ') if opt_pass else ''}batch {B:2d} plan {'in-flight' if conc else 'one-batch'}: {args.iters} steps beside the hog, "
                        f"{bad} mismatching steps, {dt * 1e3:.0f} ms; new instantiations: " + "; ".join(f"{k} x{g}" for k, g in new))
            seen_all |= {(name, opt_pass) + k for k in seen}
            net.CleanUp()
    chk = getattr(_lib.lib(), "tf2_check_dma_errors", None)
    if chk is not None:
        import ctypes as C
        v = (C.c_ulonglong * 2)()
        chk.argtypes = [C.POINTER(C.c_ulonglong)]
        chk.restype = C.c_int
        chk(v)
        say(f"# TF2_CHECK_DMA build: {v[1]} fragment reads checked, {v[0]} saw the sentinel of a chunk whose DMA had not landed")
        total_bad += int(v[0])
    say(f"# {total_cases} cases, {len(seen_all)} (kernel instantiation, grid) pairs, {total_bad} failures")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
