"""SSD300 around the integer engine.

The convolutional part of SSD300-VGG runs as one TF2 table program (`config.ssd300_tables`: VGG base with the
ceil-mode pool, stride-1 pool5, dilated conv6, conv7, the extras and the twelve multibox head convolutions) on the
integer kernels.  What the reference keeps in float on the host side of the detector is restated here with PyTorch
(GPU when available) and pinned to the reference's own functions executed in the build container
(tests/golden/ref_ssd.npz, oracle/gen_golden.py gen_ssd):

  prior boxes   TransForm_Kit/Quantization/models/SSD/layers/functions/prior_box.py:28-57
  decode        .../layers/box_utils.py:140-158
  nms           .../layers/box_utils.py:175-239   (greedy, top_k highest scores first)
  Detect        .../layers/functions/detection.py:27-62
  L2Norm        .../layers/modules/l2norm.py:19-24

The VOC / COCO prior-box configurations are the data of TransForm_Kit/Quantization/data/SSD/config.py:15-45.
No trained SSD weights or datasets ship with the reference, so the mAP of its README cannot be re-measured here.
"""
from __future__ import annotations

from math import sqrt
from typing import Dict, List, Sequence, Tuple

import numpy as np

VOC = dict(num_classes=21, feature_maps=[38, 19, 10, 5, 3, 1], min_dim=300, steps=[8, 16, 32, 64, 100, 300],
           min_sizes=[30, 60, 111, 162, 213, 264], max_sizes=[60, 111, 162, 213, 264, 315],
           aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]], variance=[0.1, 0.2], clip=True, name="VOC")
COCO = dict(num_classes=201, feature_maps=[38, 19, 10, 5, 3, 1], min_dim=300, steps=[8, 16, 32, 64, 100, 300],
            min_sizes=[21, 45, 99, 153, 207, 261], max_sizes=[45, 99, 153, 207, 261, 315],
            aspect_ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]], variance=[0.1, 0.2], clip=True, name="COCO")


def prior_boxes(cfg: dict = VOC):
    """Default boxes in centre form, [sum_k f_k^2 * boxes_k, 4] (8732 rows for SSD300), in the reference's order
    (prior_box.py:28-57): per source map k, per cell (row i, column j): the min-size square, the sqrt(min*max) square,
    then for every aspect ratio a the (w, h) pairs (s*sqrt(a), s/sqrt(a)) and (s/sqrt(a), s*sqrt(a)); clipped to
    [0, 1] when cfg['clip'].  Computed per map as a grid of centres times a small table of shapes, in float64 like the
    reference's Python arithmetic, then stored as float32."""
    import torch
    size = float(cfg["min_dim"])
    blocks = []
    for f, step, smin, smax, ratios in zip(cfg["feature_maps"], cfg["steps"], cfg["min_sizes"], cfg["max_sizes"],
                                            cfg["aspect_ratios"]):
        cells = size / step                                   # the reference divides by image_size / step
        centres = (np.arange(f, dtype=np.float64) + 0.5) / cells
        cy, cx = np.meshgrid(centres, centres, indexing="ij")
        s = smin / size
        shapes = [(s, s), (sqrt(s * (smax / size)),) * 2]
        for a_r in ratios:
            shapes += [(s * sqrt(a_r), s / sqrt(a_r)), (s / sqrt(a_r), s * sqrt(a_r))]
        wh = np.asarray(shapes, np.float64)                   # [boxes_k, 2]
        grid = np.stack([cx, cy], -1).reshape(f * f, 1, 2)    # [cells, 1, (cx, cy)]
        blocks.append(np.concatenate([np.broadcast_to(grid, (f * f, len(shapes), 2)),
                                      np.broadcast_to(wh[None], (f * f, len(shapes), 2))], -1).reshape(-1, 4))
    out = torch.from_numpy(np.concatenate(blocks, 0).astype(np.float32))
    return out.clamp_(0, 1) if cfg["clip"] else out


def decode(loc, priors, variances: Sequence[float]):
    """box_utils.py:140-158: offsets + centre-form priors -> corner-form boxes."""
    import torch
    boxes = torch.cat((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                       priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def nms(boxes, scores, overlap: float = 0.5, top_k: int = 200):
    """box_utils.py:175-239: indices of the kept boxes, best first; only the top_k scores are considered; a box
    survives a kept one when IoU <= overlap."""
    import torch
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1) * (y2 - y1)
    _, idx = scores.sort(0)
    idx = idx[-top_k:]
    keep: List[int] = []
    while idx.numel() > 0:
        i = idx[-1]
        keep.append(int(i))
        if idx.numel() == 1:
            break
        idx = idx[:-1]
        xx1 = torch.clamp(x1[idx], min=float(x1[i])); yy1 = torch.clamp(y1[idx], min=float(y1[i]))
        xx2 = torch.clamp(x2[idx], max=float(x2[i])); yy2 = torch.clamp(y2[idx], max=float(y2[i]))
        inter = torch.clamp(xx2 - xx1, min=0.0) * torch.clamp(yy2 - yy1, min=0.0)
        union = (area[idx] - inter) + area[i]
        idx = idx[(inter / union).le(overlap)]
    return torch.tensor(keep, dtype=torch.long, device=boxes.device)


def l2norm(x, weight, eps: float = 1e-10):
    """l2norm.py:19-24: x / (||x||_2 over channels + eps) * weight[c]."""
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + eps
    return weight.view(1, -1, 1, 1) * (x / norm)


def detect(loc, conf, priors, num_classes: int, top_k: int = 200, conf_thresh: float = 0.01, nms_thresh: float = 0.45,
           variance: Sequence[float] = (0.1, 0.2)):
    """detection.py:27-62: loc [B, P, 4], conf [B, P, num_classes] (already softmaxed) -> [B, num_classes, top_k, 5]
    rows (score, x1, y1, x2, y2), class 0 = background left empty."""
    import torch
    num = loc.size(0)
    output = torch.zeros(num, num_classes, top_k, 5, device=loc.device)
    conf_preds = conf.transpose(2, 1)
    for i in range(num):
        decoded = decode(loc[i], priors, variance)
        for cl in range(1, num_classes):
            c_mask = conf_preds[i, cl].gt(conf_thresh)
            scores = conf_preds[i, cl][c_mask]
            if scores.numel() == 0:
                continue
            boxes = decoded[c_mask]
            ids = nms(boxes, scores, nms_thresh, top_k)
            output[i, cl, :ids.numel()] = torch.cat((scores[ids].unsqueeze(1), boxes[ids]), 1)
    return output


def head_rows(plan) -> List[Tuple[int, int]]:
    """(loc row, conf row) per source map of a `config.ssd300_tables` program: the last twelve rows."""
    n = len(plan)
    return [(n - 12 + 2 * i, n - 12 + 2 * i + 1) for i in range(6)]


def gather_heads(read_layer, plan, q_rows: Dict[int, np.ndarray], batch: int, num_classes: int):
    """Engine outputs -> (loc [B, P, 4], conf [B, P, num_classes]) float32 as SSD.forward builds them
    (SSD.py:62-70: permute(0, 2, 3, 1), flatten, concatenate over the six sources).  read_layer(l) returns the int8
    NCHW output of row l; q_rows[l] is that row's file Q (value = int8 / 2^Q)."""
    import torch
    locs, confs = [], []
    for lrow, crow in head_rows(plan):
        for row, dst in ((lrow, locs), (crow, confs)):
            y = np.asarray(read_layer(row), np.float32) / np.exp2(np.asarray(q_rows[row], np.float32))[None, :, None, None]
            dst.append(torch.from_numpy(y).permute(0, 2, 3, 1).contiguous().view(batch, -1))
    loc = torch.cat(locs, 1).view(batch, -1, 4)
    conf = torch.cat(confs, 1).view(batch, -1, num_classes)
    return loc, conf
