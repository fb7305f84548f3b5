"""Seeded synthetic inputs in the reference's own file formats.

The trained weight files (param.bin / fpgamodel.bin) are not part of the reference
repository (Runtime_Engine/cnn/host/model/README:1-7), so parity and benchmarks use
INQ-like synthetic weights written in the exact LoadModel stream order
(model_loader.cpp:139-213; writer caffe2fpga.cpp:91-113): per layer filters
[N][C][k][k] float32, then [bias], then [mean, variance, scale_factor(1), gamma, beta].

Weights follow the INQ statistics the reference documents (TransForm_Kit/Compression/
compress_net/core/compress_train_eval.py:54, 4bit_data_format.txt): every layer has 7
magnitudes 2^e_max ... 2^(e_max-6) plus zero, random sign, ~10 % zeros.  BN statistics
are chosen so that activations stay inside the int8 range without saturating
everywhere (variance = the conv output's expected variance).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import config as cfg


def synth_q_values(tables: cfg.NetTables, seed: int = 0, lo: int = 2, hi: int = 5, spread: int = 1) -> np.ndarray:
    """Q-file ints (file order, quantization.cpp:36-53) for networks without a shipped Q
    file: per tensor a base Q in [lo,hi] with per-channel jitter of `spread`; tensors that
    are added together (residual) share one Q vector, as in the shipped resnet50_Q."""
    rng = np.random.default_rng(seed)
    plan = cfg.build_plan(tables)
    rows: Dict[int, np.ndarray] = {}
    vals: List[int] = [2, 2, 2][:3]
    for L in plan:
        if L.ipool == 1:
            continue
        if L.ipool == 2:                     # L2Norm output: weight ~ 20 times a unit vector -> Q = 2 keeps it in int8
            rows[L.index] = np.full(L.N, 2) - rng.integers(0, 2, size=L.N)
            continue
        if L.add_src >= 0 and L.add_src in rows and rows[L.add_src].size == L.N:
            r = rows[L.add_src]
        else:
            base = int(rng.integers(lo, hi + 1))
            r = np.clip(base - rng.integers(0, spread + 1, size=L.N), 0, 7)
            if L.index == len(plan) - 1:
                r = np.full(L.N, base)
        rows[L.index] = r
    # a residual source must carry the same Q as the layer that adds onto it
    for L in plan:
        if L.add_src >= 0 and not plan[L.add_src].ipool:
            rows[L.add_src] = rows[L.index]
    for L in plan:
        if L.ipool != 1:
            vals.extend(int(v) for v in rows[L.index])
    return np.asarray(vals, np.int32)


def q_text(vals) -> bytes:
    return ("\n".join(str(int(v)) for v in vals) + "\n").encode()


def synth_model(tables: cfg.NetTables, q_vals, seed: int = 0, zero_frac: float = 0.1,
                dtype=np.float32) -> np.ndarray:
    """float32 model stream in LoadModel order."""
    rng = np.random.default_rng(seed + 1000)
    plan = cfg.build_plan(tables)
    out: List[np.ndarray] = []
    for L in plan:
        fan_in = L.model_C * L.model_k * L.model_k
        if not L.ipool:
            e_max = -int(rng.integers(1, 5))                     # 2^-1 .. 2^-4
            if not L.bn_en:
                # no BN to normalise: scale the weights so the real-unit output RMS is ~1
                rin0 = 25.0 if L.src == -1 else 1.0
                e_max = int(np.clip(np.round(-0.5 * np.log2(0.19 * fan_in * rin0 * rin0)), -8, -1))
            lev = rng.integers(0, 7, size=(L.N, fan_in))
            mag = np.ldexp(1.0, e_max - lev).astype(np.float32)
            sign = np.where(rng.random((L.N, fan_in)) < 0.5, -1.0, 1.0).astype(np.float32)
            w = mag * sign
            w[rng.random((L.N, fan_in)) < zero_frac] = 0.0
            out.append(w.ravel())
            row_energy = (w.astype(np.float64) ** 2).sum(axis=1)
        else:
            row_energy = np.ones(L.N)
            if L.ipool == 2:                 # L2Norm scale weights (SSD.py:24 initialises them to 20)
                out.append(rng.uniform(12.0, 24.0, size=L.N).astype(np.float32))
        if L.bias_en:
            out.append(rng.uniform(-0.5, 0.5, size=L.N).astype(np.float32))
        if L.bn_en:
            # real-unit input RMS: ~25 for the image layer (8-bit image data), ~1 elsewhere
            rin = 25.0 if L.src == -1 else 1.0
            var = np.maximum(row_energy * rin * rin, 1e-3) * rng.uniform(0.7, 1.4, size=L.N)
            mean = rng.normal(0, 0.05, size=L.N) * np.sqrt(var)
            gamma = rng.uniform(0.5, 1.5, size=L.N)
            beta = rng.uniform(-0.5, 0.5, size=L.N)
            out.append(mean.astype(np.float32)); out.append(var.astype(np.float32))
            out.append(np.asarray([1.0], np.float32))
            out.append(gamma.astype(np.float32)); out.append(beta.astype(np.float32))
    model = np.concatenate(out).astype(dtype)
    assert model.size == cfg.model_float_count(tables)
    return model


def synth_images(tables: cfg.NetTables, batch: int, seed: int = 0, kind: str = "float") -> np.ndarray:
    """Preprocessed CHW images like the shipped resnet50_data_label_100.bin (mean-subtracted
    0..255 data, range about -126..154).  kind="int8": already quantised, uniform over the
    whole int8 range (exercises the -128 negate quirk)."""
    rng = np.random.default_rng(seed + 2000)
    C, H, W = int(tables["INPUT_IMAGE_C"]), int(tables["INPUT_IMAGE_H"]), int(tables["INPUT_IMAGE_W"])
    if kind == "int8":
        return rng.integers(-128, 128, size=(batch, C, H, W)).astype(np.int8)
    x = rng.normal(0.0, 45.0, size=(batch, C, H, W))
    return np.clip(x, -126.0, 154.0).astype(np.float32)


def squeezenet_seeded_stream(seed=5):
    """Seeded float parameters of SqueezeNet 1.1 in table order (the float32 LoadModel stream of
    cfg.squeezenet11_tables, NOT power-of-two weights: this is the float model the calibrator sees).  numpy only:
    oracle/gen_golden.py loads the same numbers into the reference's PyTorch model, tests rebuild them from the seed.
    Returns (stream, per-row dict)."""
    t = cfg.squeezenet11_tables()
    rng = np.random.default_rng(seed)
    rows, parts = [], []
    for L in cfg.build_plan(t):
        fan = L.model_C * L.model_k * L.model_k
        r = dict(w=(rng.standard_normal((L.N, L.model_C, L.model_k, L.model_k)) * np.sqrt(2.0 / fan)).astype(np.float32))
        parts.append(r["w"].ravel())
        if L.bias_en:
            r["b"] = (rng.standard_normal(L.N) * 0.1).astype(np.float32); parts.append(r["b"])
        if L.bn_en:
            r["mean"] = (rng.standard_normal(L.N) * 0.2).astype(np.float32)
            r["var"] = rng.uniform(0.5, 2.0, L.N).astype(np.float32)
            r["gamma"] = rng.uniform(0.5, 1.5, L.N).astype(np.float32)
            r["beta"] = (rng.standard_normal(L.N) * 0.3).astype(np.float32)
            parts += [r["mean"], r["var"], np.ones(1, np.float32), r["gamma"], r["beta"]]
        rows.append(r)
    return np.concatenate(parts).astype(np.float32), rows



def squeezenet_calibration_images(seed: int = 17) -> np.ndarray:
    """Three seeded 1x3x227x227 float images (BASELINE configs[0] shape), [3, 1, 3, 227, 227]."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((3, 1, 3, 227, 227)) * 50.0).astype(np.float32)


def bench_network(name: str):
    """The BASELINE.json networks as bench.py, tools/steps_only.py, tools/pmc_summary.py and tools/dma_stress.py run them:
    (tables, Q values, model seed, display name, note).  resnet50: the shipped resnet50_Q + seeded INQ weights; the others: table
    programs of tf2_amd.config with synthetic per-channel Q values (spread 1) and seeded INQ weights."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if name == "resnet50":
        t = cfg.resnet50_tables()
        qv = np.loadtxt(os.path.join(root, "tests", "golden", "resnet50_Q"), dtype=np.int32)
        return t, qv, 0, "ResNet50", "54-layer TF2 table program, shipped resnet50_Q, seeded INQ weights"
    if name in ("googlenet", "resnet50_pruned"):
        # the reference's other two shipped networks (cnn.h:29-35: compile-time selection of googlenet.h / resnet50_pruned.h): the tables
        # dumped from the shipped headers and the shipped Q files (tests/golden/, oracle/gen_golden.py), seeded INQ weights
        import json
        g = os.path.join(root, "tests", "golden")
        t = cfg.NetTables(json.load(open(os.path.join(g, f"tables_{name}.json"))))
        t.setdefault("xConv1Rewrite", 1)          # (the headers describe conv1 in its executed 3x3 form over the space-to-depth image, model_loader.cpp:244-257)
        qv = np.loadtxt(os.path.join(g, f"{name}_Q"), dtype=np.int32)
        disp = {"googlenet": "GoogLeNet", "resnet50_pruned": "pruned ResNet50"}[name]
        return t, qv, 0, disp, f"the reference's shipped {name}.h table program and {name}_Q, seeded INQ weights"
    mk, disp, seed = {"squeezenet": (cfg.squeezenet11_tables, "SqueezeNet 1.1", 6), "vgg16": (cfg.vgg16_tables, "VGG16", 1),
                      "ssd300": (cfg.ssd300_tables, "SSD300-VGG", 3)}[name]
    t = mk()
    return t, synth_q_values(t, seed, spread=1), seed, disp, "TF2 table program built by tf2_amd.config, synthetic per-channel Q values and seeded INQ weights"
