"""Channel-wise activation quantisation (CAQ) calibrator -- produces TF2 Q files.

Reference: `TransForm_Kit/Quantization/feature_write.py:73-113` (float forward over calibration images keeping the
element-wise maximum of |feature| per layer) and `TransForm_Kit/Quantization/quantization.py:33-72`
(`QuantizeForShift`: the largest power-of-two scale that keeps a channel inside int8; `QuantizeChannel`: the
mean-clamp across the channels of a tensor).  The reference runs torchvision models from ImageNet folders on
CUDA; here the float forward executes the SAME table program the integer engine runs (conv / BN / ReLU / pool /
residual / global average / FC from the k* tables and the float32 LoadModel stream) with PyTorch on the GPU, so a
Q file can be produced for any network the engine accepts, from any image source.

The two numeric rules are pinned to the reference's functions executed in the build container
(tests/golden/ref_caq.npz, oracle/gen_golden.py gen_caq).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from . import config as cfg
from .model4bit import split_model


# ---- the two rules (quantization.py:33-72) -----------------------------------------------------------------
def quantize_for_shift(x: np.ndarray) -> float:
    """Largest power-of-two exponent Q with x * 2^Q inside [-128, 127] (quantization.py:33-46).

    The reference starts from log2(127 / max|x|) -- floored when negative, numpy-rounded (half to even) otherwise --
    and steps down while the scaled data overflow.  Stated directly: the start value r, then the first Q <= r whose
    scaled extremes fit; an all-zero channel keeps Q = 0."""
    v = np.asarray(x, np.float64)
    peak = float(np.abs(v).max()) if v.size else 0.0
    if peak <= 0.0:
        return 0.0
    r = np.log2(127.0 / peak)
    q = float(np.floor(r)) if r < 0 else float(np.round(r))
    hi, lo = float(v.max()), float(v.min())
    while hi * 2.0 ** q > 127 or lo * 2.0 ** q < -128:
        q -= 1.0
    return q


def quantize_channels(x: np.ndarray) -> np.ndarray:
    """Per-channel Q of a feature tensor plus the reference's mean clamp (quantization.py:48-72, style 'shift').

    Channels are axis 1 of an N-D tensor, or the elements of a vector.  After the per-channel rule, positive values
    above the mean of all channels are pulled down to floor(mean), negative values below it up to ceil(mean)."""
    v = np.asarray(x, np.float64)
    per_channel = [v[i] for i in range(v.shape[0])] if v.ndim == 1 else [v[:, i] for i in range(v.shape[1])]
    q = np.array([quantize_for_shift(c) for c in per_channel])
    m = q.mean()
    q = np.where((q > 0) & (q > m), float(math.floor(m)), q)
    q = np.where((q < 0) & (q < m), float(math.ceil(m)), q)
    return q


# ---- float forward of the table program -----------------------------------------------------------------------
def float_forward(tables: cfg.NetTables, model: np.ndarray, images, device: Optional[str] = None, bn_eps: float = 1e-5) -> Dict[int, "torch.Tensor"]:
    """float32 forward of the k*-table program: {-1: image, l: output tensor of table row l (after pool / add /
    global average, i.e. the tensor whose Q row is l + 1)}.  images: [B, C, H, W] float (numpy or torch).
    bn_eps: 1e-5 is what the runtime folds with whatever the model says (model_loader.cpp:221);
    pass the framework's value (SqueezeNet.py:23 uses 1e-3) to reproduce a calibration run of the float model."""
    import torch
    import torch.nn.functional as F
    if device is None:
        device = "cuda:0" if torch.cuda.is_available() else "cpu"
    x0 = torch.as_tensor(np.asarray(images, np.float32) if not torch.is_tensor(images) else images, dtype=torch.float32, device=device)
    plan = cfg.build_plan(tables)
    params = {name: torch.from_numpy(np.ascontiguousarray(arr)).to(device) for name, arr, _ in split_model(tables, model)}
    outs: Dict[int, torch.Tensor] = {-1: x0}
    concat: Dict[int, List] = {}

    def source(L):
        if L.src == -1:
            return x0
        if L.src >= 0:
            return outs[L.src]
        parts = sorted(concat[-(L.src + 2)], key=lambda p: p[0])
        return torch.cat([p[1] for p in parts], 1)

    for L in plan:
        x = source(L)
        if L.ipool == 2:            # L2Norm row (l2norm.py:19-24)
            wl = params[f"layer{L.index}.l2w"].reshape(1, -1, 1, 1)
            y = wl * (x / (x.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10))
        elif L.ipool:
            y = x
        else:
            w = params[f"layer{L.index}.filter"]
            if L.model_k != L.k:          # conv1 in its file form (7x7 stride 2 pad 3 over the raw image)
                y = F.conv2d(x, w, None, stride=2, padding=3)
            else:
                y = F.conv2d(x, w, None, stride=L.stride, padding=(L.pad_h, L.pad_w), dilation=L.dil)
            if L.bias_en:
                y = y + params[f"layer{L.index}.bias"].reshape(1, -1, 1, 1)
            if L.bn_en:
                sf = params[f"layer{L.index}.scale_factor"].reshape(())
                mean = params[f"layer{L.index}.mean"].reshape(1, -1, 1, 1) / sf
                var = params[f"layer{L.index}.var"].reshape(1, -1, 1, 1) / sf
                y = params[f"layer{L.index}.gamma"].reshape(1, -1, 1, 1) * (y - mean) / torch.sqrt(var + bn_eps) + \
                    params[f"layer{L.index}.beta"].reshape(1, -1, 1, 1)
            if L.relu:
                y = torch.relu(y)
        if L.pool_en:
            # zero-extended max pool (pool.cl:152-260): out-of-range taps are 0, not -inf
            y = F.max_pool2d(F.pad(y, (L.pool_pad, L.pool_S, L.pool_pad, L.pool_S)), L.pool_S, L.pool_st)[:, :, :L.PH, :L.PW]
        if L.add_src >= 0:
            y = y + outs[L.add_src]
            if L.add_relu:
                y = torch.relu(y)
        if L.endpool:
            y = y.mean(dim=(2, 3), keepdim=True)
        outs[L.index] = y
        if L.concat >= 0:
            concat.setdefault(L.concat, []).append((L.n_start, y))
    return outs


class Calibrator:
    """feature_write.py:73-113: running element-wise max of |feature| over calibration batches, then the Q rows."""

    def __init__(self, tables: cfg.NetTables, model: np.ndarray, device: Optional[str] = None, bn_eps: float = 1e-5):
        self.tables, self.model, self.device, self.bn_eps = tables, model, device, bn_eps
        self.plan = cfg.build_plan(tables)
        self.maxabs: Dict[int, "torch.Tensor"] = {}

    def observe(self, images) -> None:
        import torch
        outs = float_forward(self.tables, self.model, images, self.device, self.bn_eps)
        for k, v in outs.items():
            m = v.abs().amax(dim=0, keepdim=True)                 # max over the batch: [1, C, H, W]
            self.maxabs[k] = m if k not in self.maxabs else torch.maximum(self.maxabs[k], m)

    def q_rows(self) -> Dict[int, np.ndarray]:
        """{-1: image row, l: row of conv l} as the reference writes them (non-negated ints)."""
        rows = {}
        for k, v in self.maxabs.items():
            if k >= 0 and self.plan[k].ipool == 1:
                continue
            rows[k] = quantize_channels(v.detach().cpu().numpy().astype(np.float64)).astype(np.int64)
        # tensors that are added together must share one Q vector (feature_writer.cl:119-122 adds raw int8 values;
        # the shipped resnet50_Q has identical rows for every residual pair): take the element-wise minimum
        for L in self.plan:
            if L.add_src >= 0 and L.add_src in rows and rows[L.add_src].size == rows[L.index].size:
                m = np.minimum(rows[L.index], rows[L.add_src])
                rows[L.index] = m.copy(); rows[L.add_src] = m.copy()
        return rows

    def q_file_text(self) -> str:
        """The combined Q file (quantization.py:146-152 / quantization.cpp:36-53): image row, then one line per
        output channel of every conv row, in table order."""
        rows = self.q_rows()
        vals: List[int] = [int(v) for v in rows[-1]]
        for L in self.plan:
            if L.ipool != 1:
                vals += [int(v) for v in rows[L.index]]
        assert len(vals) == cfg.q_value_count(self.tables)
        return "".join(f"{v}\n" for v in vals)
