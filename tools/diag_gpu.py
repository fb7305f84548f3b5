#!/usr/bin/env python3
"""GPU bring-up diagnostic: runs small nets in the three kernel selections against the oracle
and writes WHERE mismatches are (layer, channel/pixel pattern) to gpurun_out/diag.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import netref
from tf2_amd import config as cfg, network, synth


def diag(name, t, q, model, x, mode):
    rec = dict(name=name, mode=mode, layers=[])
    try:
        net = network.NetWork(t)
        net.Init(model, synth.q_text(q), device="cuda:0", pack_mode=mode)
        r = network.Runner(None, net)
        lg = r.run_batch(torch.from_numpy(x).to("cuda:0"), keep_all=True)
        torch.cuda.synchronize()
        ref = netref.RefNet(t, q, model)
        outs = ref.run(x)
        B = x.shape[0]
        for li in [-1] + [L.index for L in ref.plan]:
            got = r.read_layer(li, B)
            want = outs[li]
            bad = np.argwhere(got != want)
            e = dict(layer=li, shape=list(want.shape), n_bad=int(len(bad)), n=int(want.size))
            if len(bad):
                e["first"] = [[int(v) for v in b] + [int(got[tuple(b)]), int(want[tuple(b)])] for b in bad[:12]]
                e["bad_channels"] = sorted(set(int(b[1]) for b in bad))[:64]
                e["bad_batches"] = sorted(set(int(b[0]) for b in bad))
                if want.ndim == 4:
                    e["bad_hw"] = sorted(set((int(b[2]), int(b[3])) for b in bad))[:32]
            rec["layers"].append(e)
        rec["logits_ok"] = bool((lg.cpu().numpy() == ref.logits(outs)).all())
    except Exception as ex:  # noqa
        rec["error"] = repr(ex)
    return rec


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = []
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 5, spread=2)
    model = synth.synth_model(t, q, 5)
    x = synth.synth_images(t, 3, 5)
    for mode in (2, 0, 1):
        out.append(diag("tiny", t, q, model, x, mode))
    t = cfg.tiny_tables(hw=20, widths=(64, 128), classes=100)
    q = synth.synth_q_values(t, 8, spread=2)
    model = synth.synth_model(t, q, 8)
    x = synth.synth_images(t, 2, 8)
    for mode in (2, 0):
        out.append(diag("tiny64", t, q, model, x, mode))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w"), indent=1)
    for r in out:
        bad = [(e["layer"], e["n_bad"]) for e in r.get("layers", []) if e["n_bad"]]
        print(r["name"], "mode", r["mode"], "error" if "error" in r else "", r.get("error", ""), "bad layers:", bad, "logits_ok", r.get("logits_ok"))


if __name__ == "__main__":
    main()
