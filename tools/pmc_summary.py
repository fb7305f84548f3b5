#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/pmc_run.sh (gpurun_out/pmc/*) into
profiles/<name>.json: per conv launch of the LAST step -- HBM bytes (FETCH_SIZE / WRITE_SIZE,
KiB units; FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM' prescribes for wide coalesced reads on
gfx950 -- validated here against layers whose byte count is known), wave cycles and instruction mix."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", os.environ.get("PMC_DIR", "pmc"))
CONC = int(os.environ.get("PMC_CONC", "0"))      # which launch plan the profiled process ran (tools/pmc_run.sh <batch> <conc> <dir>)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_pmc_conv_b32.json")

def table(d):
    f = glob.glob(f"{src}/{d}/runc/*_counter_collection.csv")[0]
    disp = collections.OrderedDict()
    for x in csv.DictReader(open(f)):
        e = disp.setdefault(int(x["Dispatch_Id"]), {"kernel": x["Kernel_Name"], "grid": int(x["Grid_Size"]), "vgpr": int(x["VGPR_Count"])})
        e[x["Counter_Name"]] = float(x["Counter_Value"])
    return [v for v in disp.values() if "tf2::conv_" in v["kernel"] or "tf2::fc_" in v["kernel"]]

# conv launches of ONE step in launch order -> table rows (layers).  The launch list is the library's own
# (tf2_net_describe_launches: tile shapes, fused pairs, split-K variants are decided in one place); a conv_bneck launch computes
# two layers (the 3x3 and the 1x1 expand behind it): its counters go to the first, the second gets an all-zero row.
sys.path.insert(0, ROOT)
import numpy as np
from tf2_amd import config as cfg, network, synth
NET = os.environ.get("PMC_NET", "resnet50")
_t, _q, _seed, _name, _ = synth.bench_network(NET)
_net = network.NetWork(_t); _net.Quantization(synth.q_text(_q)); _net.LoadModel(synth.synth_model(_t, _q, _seed)); _net.Pack(0)
B = int(os.environ.get("PMC_BATCH", "32"))
_plan = cfg.build_plan(_t)
_launches = [l for l in _net.describe_launches(B, CONC) if l["kernel"].startswith(("conv_", "fc_"))]
launch_layers = [l["layer"] for l in _launches]
fused_into = {}
for k, l in enumerate(_launches):          # a launch that computes several table rows (conv_bneck pair, conv_mfma2 pair launch)
    nxt = launch_layers[k + 1] if k + 1 < len(launch_layers) else len(_plan)
    for j in range(l["layer"] + 1, nxt):
        if not _plan[j].ipool:
            fused_into[j] = l["layer"]
n = len(launch_layers)
sq1, sq2, t1, t2 = (table(d)[-n:] for d in ("sq1", "sq2", "tcc1", "tcc2"))
for k, l in enumerate(_launches):          # the profiler's rows are the launch plan's rows, kernel by kernel
    base = l["kernel"].split("<")[0]
    for tb in (sq1, sq2, t1, t2):
        assert base in tb[k]["kernel"], (k, l, tb[k]["kernel"])
assert NET != "resnet50" or "conv_stem" in sq1[0]["kernel"]
rows = [None] * len(_plan)
for i in fused_into:
    rows[i] = dict(layer=i, kernel="(computed by the launch of layer %d)" % fused_into[i], fused_into=fused_into[i], grid_threads=0, vgpr=0,
                   fetch_bytes=0.0, fetch_bytes_raw_counter=0.0, write_bytes=0.0, waves=0.0, wave_cycles_quad=0.0, wait_any=0.0, wait_inst_any=0.0,
                   active_inst_any=0.0, insts_valu=0.0, insts_salu=0.0, insts_lds=0.0, insts_vmem=0.0, insts_mfma=0.0, mfma_busy_cycles=0.0, lds_bank_conflict=0.0)
for k, i in enumerate(launch_layers):
    a, b, c, d = sq1[k], sq2[k], t1[k], t2[k]
    prev = rows[i] if (rows[i] is not None and "fused_into" not in rows[i]) else None      # a row with two launches (conv_fc: partial + finish): summed
    rows[i] = dict(layer=i, kernel=a["kernel"].split("(")[0][:60], grid_threads=a["grid"], vgpr=a["vgpr"],
                     fetch_bytes=2 * c["FETCH_SIZE"] * 1024, fetch_bytes_raw_counter=c["FETCH_SIZE"] * 1024,
                     write_bytes=d["WRITE_SIZE"] * 1024, waves=a["SQ_WAVES"], wave_cycles_quad=a["SQ_WAVE_CYCLES"],
                     wait_any=a["SQ_WAIT_ANY"], wait_inst_any=a["SQ_WAIT_INST_ANY"], active_inst_any=a["SQ_ACTIVE_INST_ANY"],
                     insts_valu=a["SQ_INSTS_VALU"], insts_salu=b["SQ_INSTS_SALU"], insts_lds=b["SQ_INSTS_LDS"],
                     insts_vmem=b["SQ_INSTS_VMEM"], insts_mfma=b["SQ_INSTS_MFMA"], mfma_busy_cycles=b["SQ_VALU_MFMA_BUSY_CYCLES"],
                     lds_bank_conflict=b["SQ_LDS_BANK_CONFLICT"])
    if prev is not None:
        for f in rows[i]:
            if isinstance(rows[i][f], float):
                rows[i][f] += prev[f]
        rows[i]["kernel"] = prev["kernel"] + " + " + rows[i]["kernel"]
for i in range(len(rows)):                 # rows without a conv launch of their own (independent pools, L2Norm): all-zero
    if rows[i] is None:
        rows[i] = dict(layer=i, kernel="(no conv launch: pool / L2Norm row)", grid_threads=0, vgpr=0, fetch_bytes=0.0, fetch_bytes_raw_counter=0.0, write_bytes=0.0)
tot = dict(fetch_bytes=sum(r["fetch_bytes"] for r in rows), write_bytes=sum(r["write_bytes"] for r in rows))
json.dump(dict(note="rocprofv3 --pmc passes (separate runs: SQ x2, FETCH_SIZE+GRBM, WRITE_SIZE), %s batch %d, one step of the %s launch plan (a launch that covers several table rows: counters on the first); " % (_name, B, "batches-in-flight" if CONC else "one-batch-at-a-time") +
                    "fetch_bytes = 2 x FETCH_SIZE KiB (gfx950 correction)", total=tot, layers=rows), open(out, "w"), indent=1)
print("wrote", out, {k: round(v / 1e6, 1) for k, v in tot.items()}, "MB per step")
