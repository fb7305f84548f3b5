#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python oracle/gen_golden.py

What it writes (all data: inputs + expected outputs, never reference source text):
  tables_<net>.json        every k* table of the shipped resnet50/googlenet/resnet50_pruned
                           headers and of the header TF2_auto_config generates from the shipped
                           fpganetwork.bin, dumped by compiling oracle/ref_dump_tables.cpp
                           against the reference headers (oracle/Makefile).
  fpganetwork_resnet50.bin, resnet50_Q, googlenet_Q, resnet50_pruned_Q,
  resnet50_data_label_100.bin, resnet50_fc1000_label_100.bin
                           the data files the reference itself ships as its test fixtures.
  ref_host.npz             outputs of the reference's own compiled host functions
                           (oracle/_ref/libtf2ref_*.so): Get_real, LoadModel (codes / BiasBnParam,
                           as CRCs + small layers in full, on the seeded synthetic model of
                           tf2_amd.synth), filter_trans, feature_trans, Quantization, Evaluation.
  ref_pyemu.npz            outputs of the reference's Python FPGA emulator functions
  ref_pyemu_block.npz      every intermediate tensor of the emulator's own Bottleneck.forward / ResNet.forward (head + tail)
                           on small seeded blocks, plus its BN on > 10^5 samples with exact rounding ties
  ref_caq.npz              outputs of the reference's calibrator functions QuantizeForShift / QuantizeChannel
  ref_caq_squeezenet.npz   Q vectors of the reference's own calibration pass (feature_hook + QuantizeChannel) over its own
                           SqueezeNet 1.1 forward at 1x3x227x227 with seeded parameters (BASELINE configs[0])
  ref_ssd.npz              outputs of the reference's SSD PriorBox / decode / nms (+ the L2Norm formula)
                           (TransForm_Kit/Quantization/debug/...Batch-2.py: Conv2dInt8, BN, FC),
                           AST-extracted and executed here.
"""
import ast
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
RE = REF + "/Runtime_Engine"
OUT = os.path.join(ROOT, "tests", "golden")
REFOUT = os.path.join(HERE, "_ref")

from tf2_amd import config as cfg, synth  # noqa: E402


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


class BiasBn(C.Structure):
    _fields_ = [("bias", C.c_int32), ("alpha", C.c_int32), ("beta", C.c_int32)]


def ref_lib(net):
    L = C.CDLL(os.path.join(REFOUT, f"libtf2ref_{net}.so"))
    L._Z8Get_realfc.restype = C.c_char
    L._Z8Get_realfc.argtypes = [C.c_float, C.c_char]
    return L


def gen_tables_and_data():
    for net in ("resnet50", "googlenet", "resnet50_pruned", "gen_resnet50"):
        shutil.copy(os.path.join(REFOUT, f"tables_{net}.json"), os.path.join(OUT, f"tables_{net}.json"))
    shutil.copy(RE + "/TF2_auto_config/examples/resnet50/fpganetwork.bin", os.path.join(OUT, "fpganetwork_resnet50.bin"))
    for f in ("resnet50_Q", "googlenet_Q", "resnet50_pruned_Q"):
        shutil.copy(RE + "/cnn/host/model/" + f, os.path.join(OUT, f))
    shutil.copy(RE + "/cnn/host/test_images/resnet50_data_label_100.bin", os.path.join(OUT, "resnet50_data_label_100.bin"))
    shutil.copy(RE + "/cnn/host/verify/resnet50_fc1000_label_100.bin", os.path.join(OUT, "resnet50_fc1000_label_100.bin"))


def gen_ref_host():
    out = {}
    L = ref_lib("resnet50")
    # ---- Get_real ------------------------------------------------------------------
    rng = np.random.default_rng(7)
    vals = [0.0, 1e-6, -1e-6, 9.9e-6, 1.01e-5, 0.3, -0.3, 1.5, -2.0, 3.0, 0.75, 1e-3]
    for i in range(0, 17):
        for s in (1.0, -1.0):
            for f in (1.0, 0.9901, 0.9899, 1.0099, 1.0101, 0.995, 1.005):
                vals.append(s * f * 2.0 ** (-i))
    vals += list(rng.normal(0, 0.2, 200))
    vals = np.asarray(vals, np.float32)
    expands = np.asarray(list(range(-3, 32)) + [40, 100, 127, -128], np.int8)
    gr = np.empty((vals.size, expands.size), np.uint8)
    for i, v in enumerate(vals):
        for j, e in enumerate(expands):
            r = L._Z8Get_realfc(C.c_float(float(v)), C.c_char(int(e) & 0xff))
            gr[i, j] = r[0] if isinstance(r, bytes) else (int(r) & 0xff)
    out.update(getreal_vals=vals, getreal_expands=expands, getreal_codes=gr)

    # ---- Quantization + LoadModel on the seeded synthetic ResNet50 model -----------------
    tables = cfg.parse_net_header(RE + "/cnn/host/inc/resnet50.h")
    plan = cfg.build_plan(tables)
    NL, MAXC = 54, 2048
    q = np.zeros((55, MAXC), np.int8)
    dummy = np.zeros(16, np.float32)
    qfile = (RE + "/cnn/host/model/resnet50_Q").encode()
    L._Z12QuantizationPcPfS_(q.ctypes.data_as(C.c_void_p), dummy.ctypes.data_as(C.c_void_p), C.c_char_p(qfile))
    out.update(q_resnet50_crc=np.asarray([crc(q)], np.uint32), q_resnet50_rows=q[[0, 1, 2, 5, 44, 54]].copy())
    qv = np.loadtxt(RE + "/cnn/host/model/resnet50_Q", dtype=np.int32)
    model = synth.synth_model(tables, qv, seed=0)
    MAX_FILTER = 262144 * 64
    filt = np.full(NL * MAX_FILTER, 0x40, np.uint8)
    bb = (BiasBn * (NL * 2048))()
    with tempfile.TemporaryDirectory() as td:
        mp = os.path.join(td, "model.bin")
        model.tofile(mp)
        L._Z9LoadModelPcS_P11BiasBnParamS_(C.c_char_p(mp.encode()), filt.ctypes.data_as(C.c_void_p), bb, q.ctypes.data_as(C.c_void_p))
    bba = np.frombuffer(bb, dtype=np.int32).reshape(NL, 2048, 3)
    codes_crc, bias_crc, alpha_crc, beta_crc = [], [], [], []
    small = {}
    for l, Lp in enumerate(plan):
        n_codes = Lp.N * Lp.C * Lp.k * Lp.k
        c = filt[l * MAX_FILTER: l * MAX_FILTER + n_codes]
        codes_crc.append(crc(c))
        bias_crc.append(crc(bba[l, :Lp.N, 0])); alpha_crc.append(crc(bba[l, :Lp.N, 1])); beta_crc.append(crc(bba[l, :Lp.N, 2]))
        if l in (0, 2, 53):
            small[f"lm_codes_{l}"] = c.copy() if l != 53 else c[:64 * 2048].copy()
            small[f"lm_bias_{l}"] = bba[l, :Lp.N, 0].copy(); small[f"lm_alpha_{l}"] = bba[l, :Lp.N, 1].copy()
            small[f"lm_beta_{l}"] = bba[l, :Lp.N, 2].copy()
    out.update(lm_seed=np.asarray([0]), lm_codes_crc=np.asarray(codes_crc, np.uint32), lm_bias_crc=np.asarray(bias_crc, np.uint32),
               lm_alpha_crc=np.asarray(alpha_crc, np.uint32), lm_beta_crc=np.asarray(beta_crc, np.uint32), **small)
    del filt

    # ---- filter_trans --------------------------------------------------------------------
    planes = rng.integers(0, 256, size=(16, 49)).astype(np.uint8)
    ft = np.zeros((16, 81), np.uint8)          # LoadModel pre-clears with memset(0) (model_loader.cpp:246)
    for i in range(16):
        L._Z12filter_transPcS_(planes[i].ctypes.data_as(C.c_void_p), ft[i].ctypes.data_as(C.c_void_p))
    out.update(ft_in=planes, ft_out=ft)

    # ---- feature_trans / LoadInputImage --------------------------------------------------
    raw = np.zeros(3 * 224 * 224 * 2, np.float32)
    inp = np.zeros(27 * 114 * 114, np.float32)
    imgf = (RE + "/cnn/host/test_images/resnet50_data_label_100.bin").encode()
    L._Z14LoadInputImagePcPfS0_i(C.c_char_p(imgf), inp.ctypes.data_as(C.c_void_p), raw.ctypes.data_as(C.c_void_p), 0)
    out.update(lii_crc=np.asarray([crc(inp)], np.uint32), lii_sample_idx=np.arange(0, inp.size, 9973), lii_sample=inp[::9973].copy())
    rimg = rng.normal(0, 50, size=(224 * 224,)).astype(np.float32)
    fo = np.zeros(9 * 115 * 115 + 2048, np.float32)
    L._Z13feature_transPfS_(rimg.ctypes.data_as(C.c_void_p), fo.ctypes.data_as(C.c_void_p))
    f9 = fo[:9 * 115 * 115].reshape(9, 115, 115)[:, :114, :114]
    out.update(ftr_seed_plane=rimg, ftr_out_crc=np.asarray([crc(np.ascontiguousarray(f9))], np.uint32))

    # ---- Evaluation (top-5 with ties) ----------------------------------------------------
    ev_logits, ev_labels = [], []
    OUTPUT_OFFSET = int(tables["OUTPUT_OFFSET"])
    ddr_base = int(tables["kDDRWriteBase"][53]) * 128
    qE = np.zeros((55, MAXC), np.int8)
    qE[54, :1000] = -np.asarray(rng.integers(0, 4, 1000), np.int8)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            for case in range(6):
                lg = rng.integers(-128, 128, 1000).astype(np.int8)
                if case >= 3:
                    lg = rng.integers(-5, 6, 1000).astype(np.int8)      # many ties
                buf = np.zeros(2 * OUTPUT_OFFSET + 2048 * 8 + ddr_base + 1024, np.int8)
                for n in range(1000):
                    buf[ddr_base + OUTPUT_OFFSET + (n // 16) * 128 + (n % 16)] = lg[n]
                lab = np.zeros(5, np.int32)
                L._Z10EvaluationiPcS_Pi(0, qE.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p))
                ev_logits.append(lg); ev_labels.append(lab.copy())
        finally:
            os.chdir(cwd)
    out.update(ev_q=qE[54, :1000].copy(), ev_logits=np.stack(ev_logits), ev_labels=np.stack(ev_labels))

    # ---- GoogLeNet Quantization (concat mirroring, ipool rows) ---------------------------
    G = ref_lib("googlenet")
    gt = cfg.parse_net_header(RE + "/cnn/host/inc/googlenet.h")
    gq = np.zeros((int(gt["NUM_Q_LAYERS"]), int(gt["MAX_OUT_CHANNEL"])), np.int8)
    G._Z12QuantizationPcPfS_(gq.ctypes.data_as(C.c_void_p), dummy.ctypes.data_as(C.c_void_p),
                             C.c_char_p((RE + "/cnn/host/model/googlenet_Q").encode()))
    out.update(q_googlenet=gq)
    np.savez_compressed(os.path.join(OUT, "ref_host.npz"), **out)


def gen_pyemu():
    np.lib.pad = np.pad
    src = open(REF + "/TransForm_Kit/Quantization/debug/Pytorch-ResNet50-Log2QuantizeLoad-FPGA_Quantize-Batch-2.py").read()
    tree = ast.parse(src)
    want = {"Conv2dInt8", "BN", "FC"}
    mod = ast.Module([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], [])
    import torch
    ns = {"np": np, "torch": torch, "print": lambda *a, **k: None}
    exec(compile(mod, "ref_debug", "exec"), ns)
    rng = np.random.default_rng(11)
    out = {}
    for idx, (Cc, N, H, k, stride, pad) in enumerate([(8, 6, 9, 3, 1, 1), (16, 4, 8, 1, 1, 0), (5, 7, 11, 3, 2, 1), (3, 4, 12, 5, 1, 2)]):
        x = rng.integers(-127, 128, size=(2, Cc, H, H)).astype(np.int8)        # no -128 (Appendix C-1)
        shift = rng.integers(0, 21, size=(N, Cc, k, k)).astype(np.int32)
        sign = rng.choice(np.asarray([-1, 0, 1], np.int8), size=(N, Cc, k, k), p=[0.45, 0.1, 0.45]).astype(np.int8)
        acc = ns["Conv2dInt8"](x, shift, sign, stride, pad)
        out[f"conv{idx}_x"] = x; out[f"conv{idx}_shift"] = shift; out[f"conv{idx}_sign"] = sign
        out[f"conv{idx}_geom"] = np.asarray([stride, pad], np.int32); out[f"conv{idx}_acc"] = np.asarray(acc, np.int32)
    # BN: float emulation vs the integer requant -- keep cases away from .5 ties
    N = 64
    Qout = rng.integers(0, 6, N).astype(np.int8)
    alpha = rng.uniform(0.02, 2.0, N).astype(np.float32)
    beta = rng.uniform(-1.5, 1.5, N).astype(np.float32)
    acc = rng.integers(-(1 << 24), 1 << 24, size=(1, N, 6, 6)).astype(np.int32)
    ns.update(layer_count=0, layer_name_binQ=["k"], Q={"k": [int(v) for v in Qout]}, INFLAT=15)
    y = ns["BN"](acc, np.zeros(N, np.float32), alpha, beta)
    exact = (alpha.astype(np.float64)[None, :, None, None] * acc.astype(np.float64) +
             beta.astype(np.float64)[None, :, None, None] * 2.0 ** (Qout.astype(np.float64) + 15)[None, :, None, None]) * 2.0 ** -15
    frac = np.abs(exact - np.floor(exact) - 0.5)
    safe = (frac > 0.02) & (np.abs(exact) < 120)
    out.update(bn_acc=acc, bn_alpha=alpha, bn_beta=beta, bn_q=Qout, bn_y=np.asarray(y, np.float32), bn_safe=safe)
    # FC
    ns["BatchSize"] = 1
    xf = rng.integers(0, 128, size=(1, 96)).astype(np.int8)
    sh = rng.integers(0, 16, size=(10, 96)).astype(np.int32)
    sg = rng.choice(np.asarray([-1, 0, 1], np.int8), size=(10, 96)).astype(np.int8)
    bias = rng.integers(-(1 << 18), 1 << 18, 10).astype(np.float32)
    yf = ns["FC"](xf, sh, sg, bias).numpy()
    out.update(fc_x=xf, fc_shift=sh, fc_sign=sg, fc_bias=bias, fc_y=yf.astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "ref_pyemu.npz"), **out)


def gen_pyemu_block():
    """Executes the reference emulator's OWN forward code (…Batch-2.py: class Bottleneck :231-323, class ResNet
    :325-443 with the conv1 -> BN -> clamp -> ReLU -> MaxPool2d(3,2,1) -> ... -> AdaptiveAvgPool -> round -> FC head),
    AST-extracted with the script's globals supplied here, on small seeded blocks; every intermediate tensor the
    emulator hands to its FeatureWrite / BN / Conv2dInt8 is captured.  Pins the oracle's relu, max-pool, stride-2
    subsampling, int16 residual add + clamp + ReLU, global average and requant (tests/test_golden_pyemu.py)."""
    np.lib.pad = np.pad
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    src = open(REF + "/TransForm_Kit/Quantization/debug/Pytorch-ResNet50-Log2QuantizeLoad-FPGA_Quantize-Batch-2.py").read()
    tree = ast.parse(src)
    fn = {"Conv2dInt8", "BN", "FC", "GetBias", "conv3x3", "conv1x1"}
    cl = {"Bottleneck", "ResNet"}
    mod = ast.Module([n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in fn) or
                      (isinstance(n, ast.ClassDef) and n.name in cl)], [])
    cap = []                                   # (name, array) in call order
    ns = {"np": np, "torch": torch, "nn": nn, "F": F, "print": lambda *a, **k: None, "INFLAT": np.int8(15), "BatchSize": 1}
    ns["FeatureWrite"] = lambda name, x: cap.append((name, np.array(x, np.float32, copy=True)))
    ns["FeatureWriteFC"] = lambda name, x: cap.append((name, np.array(x, np.float32, copy=True)))
    exec(compile(mod, "ref_debug_block", "exec"), ns)
    conv_raw, bn_raw = ns["Conv2dInt8"], ns["BN"]

    def conv_cap(x, w, w2, stride, padding):
        acc = conv_raw(x, w, w2, stride, padding)
        cap.append(("conv_in", np.array(x, np.int8, copy=True))); cap.append(("conv_acc", np.array(acc, np.int32, copy=True)))
        return acc

    def bn_cap(acc, bias_power, al, be):
        y = bn_raw(acc, bias_power, al, be)
        cap.append(("bn_out", np.array(y, np.float32, copy=True)))
        return y
    ns["Conv2dInt8"], ns["BN"] = conv_cap, bn_cap
    rng = np.random.default_rng(29)
    out = {}

    def weights(N, Cc, k, lo=7, hi=13):
        shift = rng.integers(lo, hi + 1, size=(N, Cc, k, k)).astype(np.int32)
        sign = rng.choice(np.asarray([-1, 0, 1], np.int8), size=(N, Cc, k, k), p=[0.45, 0.1, 0.45]).astype(np.int8)
        return shift, sign

    def bn_params(N, scale):
        # alpha on a 2^-12 grid and beta on a 2^-8 grid: the emulator's float arithmetic is then exact, so rounding
        # ties can be enumerated exactly (the FPGA rule rounds them up, pe.cl:191-193; the emulator half-to-even)
        al = (rng.integers(1, 1 << 12, N) / 4096.0 * scale).astype(np.float32)
        al = (np.round(al * 4096) / 4096).astype(np.float32); al[al == 0] = 1.0 / 4096
        be = (rng.integers(-256, 257, N) / 256.0).astype(np.float32)
        return al, be

    # ---- two bottlenecks: stride 1 without a projection, stride 2 with one -----------------------------------
    for tag, (inpl, planes, stride, H, down) in {"b1": (32, 8, 1, 9, False), "b2": (24, 8, 2, 10, True)}.items():
        width, outc = planes, planes * 4
        blk = ns["Bottleneck"](inpl, planes, stride, downsample=(object() if down else None))
        shapes = [(width, inpl, 1), (width, width, 3), (outc, width, 1)] + ([(outc, inpl, 1)] if down else [])
        names = [f"c{i}" for i in range(len(shapes))]
        Qk = [f"q{i}" for i in range(4)]
        Qv = {k: [int(v) for v in rng.integers(0, 5, 64)] for k in Qk}
        Filter, Filter2, alpha, beta = {}, {}, {}, {}
        for nme, (N, Cc, k) in zip(names, shapes):
            Filter[nme], Filter2[nme] = weights(N, Cc, k)
            # keep requantised values inside the int8 range most of the time: |acc| ~ 127 * sqrt(K) * 2^10
            alpha[nme], beta[nme] = bn_params(N, 2.0 ** 15 * 200 / (127 * np.sqrt(Cc * k * k) * 2.0 ** 11))
            out[f"{tag}_{nme}_shift"], out[f"{tag}_{nme}_sign"] = Filter[nme], Filter2[nme]
            out[f"{tag}_{nme}_alpha"], out[f"{tag}_{nme}_beta"] = alpha[nme], beta[nme]
        ns.update(layer_name_bin=[f"f{i}" for i in range(16)], layer_name_binQ=Qk, filter_name=names, bn_name=names,
                  layer_count=0, filter_count=0, feature_file_count=0, Filter=Filter, Filter2=Filter2, alpha=alpha, beta=beta, Q=Qv)
        x = rng.integers(0, 128, size=(2, inpl, H, H)).astype(np.int8)       # a block input is post-ReLU
        del cap[:]
        y = blk.forward(x)
        out[f"{tag}_x"] = x; out[f"{tag}_y"] = np.asarray(y.numpy(), np.float32)
        out[f"{tag}_geom"] = np.asarray([inpl, planes, stride, H, int(down)], np.int32)
        out[f"{tag}_q"] = np.asarray([Qv[k] for k in Qk], np.int32)
        seq = {}
        for nme, a in cap:
            seq.setdefault(nme, []).append(a)
        for i, a in enumerate(seq["conv_in"]): out[f"{tag}_conv{i}_in"] = a
        for i, a in enumerate(seq["conv_acc"]): out[f"{tag}_conv{i}_acc"] = a
        for i, a in enumerate(seq["bn_out"]): out[f"{tag}_bn{i}_out"] = a
        for nme, a in cap:
            if nme.startswith("f"): out[f"{tag}_feat_{nme}"] = a

    # ---- ResNet.forward head and tail with identity stages: conv1 7x7/s2/p3 -> BN -> clamp -> ReLU -> MaxPool2d(3,2,1)
    #      -> AdaptiveAvgPool2d(1) over 7x7 -> round -> FC ------------------------------------------------------
    net = ns["ResNet"](ns["Bottleneck"], [1, 1, 1, 1])
    for nme in ("layer1", "layer2", "layer3", "layer4"):
        setattr(net, nme, nn.Sequential())
    net.eval()
    Qk = [f"q{i}" for i in range(51)]
    Qv = {k: [int(v) for v in rng.integers(0, 4, 1000)] for k in Qk}
    Filter, Filter2, alpha, beta = {}, {}, {}, {}
    Filter["c0"], Filter2["c0"] = weights(64, 3, 7, 8, 13)
    alpha["c0"], beta["c0"] = bn_params(64, 2.0 ** 15 * 50 / (127 * np.sqrt(147) * 2.0 ** 11))
    beta["c0"] = np.abs(beta["c0"])            # mostly positive maps: the pool sees interesting values
    sh, sg = weights(1000, 64, 1, 7, 12)
    Filter["c1"], Filter2["c1"] = sh.reshape(1000, 64), sg.reshape(1000, 64)
    fc_bias = (rng.integers(-64, 65, 1000) / 64.0).astype(np.float32)
    ns.update(layer_name_bin=[f"f{i}" for i in range(8)], layer_name_binQ=Qk, filter_name=["c0", "c1"], bn_name=["c0", "c1"],
              layer_count=0, filter_count=0, feature_file_count=0, Filter=Filter, Filter2=Filter2, alpha=alpha, beta=beta,
              Q=Qv, fc_bias=fc_bias, fc_weight=None, bias_power=None)
    img = rng.integers(-127, 128, size=(1, 3, 28, 28)).astype(np.int8)
    del cap[:]
    with torch.no_grad():
        net.forward(img)
    got = dict()
    for nme, a in cap:
        got.setdefault(nme, []).append(a)
    out.update(head_img=img, head_c0_shift=Filter["c0"], head_c0_sign=Filter2["c0"], head_c0_alpha=alpha["c0"], head_c0_beta=beta["c0"],
               head_q1=np.asarray(Qv["q1"][:64], np.int32), head_qfc=np.asarray(Qv["q50"], np.int32),
               head_fc_shift=Filter["c1"], head_fc_sign=Filter2["c1"], head_fc_bias=fc_bias,
               head_conv1_acc=got["conv_acc"][0], head_conv1_bn=got["bn_out"][0], head_conv1_clamped=got["f1"][0],
               head_pool1=got["pool1"][0], head_pool5=got["pool5.txt"][0], head_fc=got["fc1000.txt"][0])

    # ---- the avgpool + round step alone on many maps (exactly the two statements of ResNet.forward :415-416) -----
    xs = rng.integers(0, 128, size=(64, 48, 7, 7)).astype(np.int8)
    xs[:8] = rng.integers(-128, 128, size=(8, 48, 7, 7)).astype(np.int8)
    out["avg_x"] = xs
    out["avg_y"] = torch.round(net.avgpool(torch.Tensor(xs.astype(np.float32)))).numpy().reshape(64, 48).astype(np.float32)

    # ---- BN on > 10^5 samples, exact-arithmetic parameter grid, with rounding ties present ---------------------
    N, HW = 64, 1600
    Qout = rng.integers(0, 6, N).astype(np.int8)
    al, be = bn_params(N, 1.0)
    al[:8] = np.asarray([1.0, 0.5, 0.25, 1.5, 1.0, 0.5, 0.75, 1.0], np.float32)
    acc = (rng.integers(-(1 << 14), 1 << 14, size=(1, N, 40, 40)).astype(np.int32) << rng.integers(7, 11, size=(1, N, 1, 1))).astype(np.int32)
    ns.update(layer_count=0, layer_name_binQ=["k"], Q={"k": [int(v) for v in Qout]})
    y = bn_raw(acc, np.zeros(N, np.float32), al, be)
    out.update(bnx_acc=acc, bnx_alpha=al, bnx_beta=be, bnx_q=Qout, bnx_y=np.asarray(y, np.float32))
    np.savez_compressed(os.path.join(OUT, "ref_pyemu_block.npz"), **out)


def gen_caq():
    """Outputs of the reference's calibrator functions (TransForm_Kit/Quantization/quantization.py:33-72, executed
    here from the AST -- the module itself needs the dataset/model loaders) on seeded max-|feature| tensors."""
    src = open(REF + "/TransForm_Kit/Quantization/quantization.py").read()
    tree = ast.parse(src)
    want = {"QuantizeForShift", "QuantizeChannel"}
    mod = ast.Module([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], [])
    import math
    ns = {"np": np, "math": math}
    exec(compile(mod, "ref_caq", "exec"), ns)
    rng = np.random.default_rng(23)
    out = {}
    # single-channel rule: magnitudes across many decades, exact powers of two, tiny, zero
    mags = np.concatenate([10.0 ** rng.uniform(-4, 4, 300), 127.0 / 2.0 ** np.arange(-8, 9), 2.0 ** np.arange(-10, 10.0),
                           127.0 * 2.0 ** rng.uniform(-6, 6, 100), [0.0, 1e-30, 126.99, 127.0, 127.01, 63.5, 63.49, 254.0]])
    qs = []
    for m in mags:
        x = np.abs(rng.uniform(0, 1, size=(3, 5))) * m
        x.flat[rng.integers(0, x.size)] = m                      # the channel's max-|feature|
        qs.append(ns["QuantizeForShift"](x))
        out.setdefault("fs_x", []).append(x)
    out["fs_x"] = np.stack(out["fs_x"]).astype(np.float64); out["fs_q"] = np.asarray(qs, np.float64)
    # whole tensors: [1, C, H, W] (conv features), [1, C] (fc features) and 1-D
    for i, shape in enumerate([(1, 16, 6, 6), (1, 64, 3, 3), (1, 10), (12,), (1, 8, 4, 4)]):
        base = 10.0 ** rng.uniform(-2, 2)
        x = np.abs(rng.normal(0, 1, size=shape)) * base * 10.0 ** rng.uniform(-1.5, 1.5, size=(shape[1] if len(shape) > 1 else shape[0],)).reshape(
            (1, -1) + (1,) * (len(shape) - 2) if len(shape) > 1 else (-1,))
        if i == 4:
            x[:, 3] = 0.0                                         # an all-zero channel keeps Q = 0
        out[f"qc{i}_x"] = x.astype(np.float64)
        out[f"qc{i}_q"] = np.asarray(ns["QuantizeChannel"]("shift", x.copy()), np.float64)
    np.savez_compressed(os.path.join(OUT, "ref_caq.npz"), **out)


def gen_caq_squeezenet():
    """BASELINE configs[0]: the reference's OWN calibration pass on its OWN SqueezeNet 1.1 (models/SqueezeNet/
    SqueezeNet.py imported here with torchvision stubbed; feature_hook of feature_write.py:72-87
    and QuantizeChannel of quantization.py:48-72 executed from their AST) over three seeded 1x3x227x227 images.  The
    model carries the seeded parameters of squeezenet_seeded_stream(); the fixture holds only the expected Q vectors
    and a few float outputs -- tests/test_calibrate.py rebuilds weights and images from the seeds."""
    import types
    import torch
    for m in ["torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models", "cv2", "torchsummary"]:
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF + "/TransForm_Kit/Quantization")
    from models.SqueezeNet import SqueezeNet as RS
    net = RS.SqueezeNet("1_1").eval()
    stream, rows = synth.squeezenet_seeded_stream()
    fires = [net.fire2, net.fire3, net.fire4, net.fire5, net.fire6, net.fire7, net.fire8, net.fire9]
    convs = [(net.conv1, net.bn1)]
    for f in fires:
        convs += [(f.squeeze, f.bn1), (f.expand1x1, f.bn2), (f.expand3x3, f.bn3)]
    convs += [(net.final_conv, None), (net.fc, net.bn)]
    with torch.no_grad():
        for (cv, bn), r in zip(convs, rows):
            w = torch.from_numpy(r["w"])
            cv.weight.copy_(w.reshape(cv.weight.shape))
            if "b" in r:
                cv.bias.copy_(torch.from_numpy(r["b"]))
            if bn is not None:
                bn.running_mean.copy_(torch.from_numpy(r["mean"])); bn.running_var.copy_(torch.from_numpy(r["var"]))
                bn.weight.copy_(torch.from_numpy(r["gamma"])); bn.bias.copy_(torch.from_numpy(r["beta"]))
    # the reference's hook (feature_write.py:72-87), its globals supplied here
    src = open(REF + "/TransForm_Kit/Quantization/feature_write.py").read()
    tree = ast.parse(src)
    ns = {"np": np, "torch": torch, "print": lambda *a, **k: None}
    exec(compile(ast.Module([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "feature_hook"], []), "ref_fw", "exec"), ns)
    qsrc = ast.parse(open(REF + "/TransForm_Kit/Quantization/quantization.py").read())
    import math
    qns = {"np": np, "math": math}
    exec(compile(ast.Module([n for n in qsrc.body if isinstance(n, ast.FunctionDef) and n.name in ("QuantizeForShift", "QuantizeChannel")], []), "ref_q", "exec"), qns)
    # module outputs in hook order (what structure_hook sizes): run once to learn the shapes
    shapes, names = [], []
    hs = []
    def reg(mod):
        def hk(m, i, o):
            shapes.append(tuple(o.shape)); names.append(m)
        hs.append(mod.register_forward_hook(hk))
    net.apply(reg)
    imgs = synth.squeezenet_calibration_images()
    with torch.no_grad():
        net(torch.from_numpy(imgs[0]))
    for h in hs: h.remove()
    Features = [np.zeros((1, 3, 227, 227))] + [np.zeros(sh) for sh in shapes]
    ns.update(Features=Features, layer_name=["m%d" % i for i in range(len(Features))], layer_count=0, call_count=0, threshold=10 ** 9,
              FeatureWrite=lambda *a: None)
    with torch.no_grad():
        for im in imgs:
            x = torch.from_numpy(im)
            Features[0] = np.maximum(abs(x.numpy()), Features[0])            # feature_write.py:103
            ns["layer_count"] = 0
            ns["feature_hook"](net, x)
            ns["call_count"] += 1
    # which module output is the tensor of our table row l?  (after BN / ReLU / pool, the tensor the row's Q describes)
    mod_index = {id(m): i + 1 for i, m in enumerate(names)}
    row_mod = [net.maxpool1]
    for f in fires:
        row_mod += [f.squeeze_activation, f.expand1x1_activation, f.expand3x3_activation]
    row_mod += [net.avgpool, net.bn]
    out = {"image_q": np.asarray(qns["QuantizeChannel"]("shift", Features[0].copy()), np.float64)}
    for l, m in enumerate(row_mod):
        fe = Features[mod_index[id(m)]]
        out[f"row{l}_q"] = np.asarray(qns["QuantizeChannel"]("shift", fe.copy()), np.float64)
        if l in (25, 26):
            out[f"row{l}_maxabs"] = fe.astype(np.float32)
        if l == 0:
            out["row0_maxabs_ch0_3"] = fe[:, :4].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_caq_squeezenet.npz"), **out)


def gen_ssd():
    """Outputs of the reference's SSD post-processing functions (TransForm_Kit/Quantization/models/SSD/layers/
    functions/prior_box.py, layers/box_utils.py decode / nms, layers/modules/l2norm.py), executed from their AST."""
    import torch, warnings
    warnings.filterwarnings("ignore")
    base = REF + "/TransForm_Kit/Quantization/models/SSD/layers/"
    ns = {"torch": torch}
    exec("from math import sqrt as sqrt\nfrom itertools import product as product", ns)
    tree = ast.parse(open(base + "functions/prior_box.py").read())
    exec(compile(ast.Module([n for n in tree.body if isinstance(n, ast.ClassDef)], []), "ref_priorbox", "exec"), ns)
    tree = ast.parse(open(base + "box_utils.py").read())
    exec(compile(ast.Module([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("decode", "nms")], []), "ref_boxutils", "exec"), ns)
    cfgsrc = ast.parse(open(REF + "/TransForm_Kit/Quantization/data/SSD/config.py").read())
    cns = {"os": os}
    exec(compile(ast.Module([n for n in cfgsrc.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") in ("voc", "coco")], []), "ref_ssdcfg", "exec"), cns)
    out = {}
    pri = ns["PriorBox"](cns["voc"]).forward()
    out["priors_voc"] = pri.numpy().astype(np.float32)
    g = torch.Generator().manual_seed(5)
    loc = torch.randn(pri.shape[0], 4, generator=g) * 0.7
    out["loc"] = loc.numpy(); out["decoded"] = ns["decode"](loc, pri, cns["voc"]["variance"]).numpy()
    for i, (n, thr, topk) in enumerate([(300, 0.45, 200), (50, 0.3, 10), (1, 0.5, 5), (800, 0.45, 200)]):
        c = torch.rand(n, 2, generator=g) * 0.8
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.02
        boxes = torch.cat([c, c + wh], 1)
        scores = torch.rand(n, generator=g)
        keep, count = ns["nms"](boxes, scores, thr, topk)
        out[f"nms{i}_boxes"] = boxes.numpy(); out[f"nms{i}_scores"] = scores.numpy()
        out[f"nms{i}_par"] = np.asarray([thr, topk], np.float64); out[f"nms{i}_keep"] = keep[:count].numpy().astype(np.int64)
    x = torch.randn(2, 16, 5, 5, generator=g)
    w = torch.rand(16, generator=g) * 20
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10          # l2norm.py:20-24
    out["l2_x"] = x.numpy(); out["l2_w"] = w.numpy()
    out["l2_y"] = (w.unsqueeze(0).unsqueeze(2).unsqueeze(3).expand_as(x) * torch.div(x, norm)).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_ssd.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    gen_tables_and_data()
    gen_ref_host()
    gen_pyemu()
    gen_pyemu_block()
    gen_caq()
    gen_caq_squeezenet()
    gen_ssd()
    print("golden fixtures written to", OUT)
