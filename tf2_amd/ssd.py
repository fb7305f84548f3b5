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
    """Regression offsets -> boxes (semantics of box_utils.py:140-158).  A prior is (cx, cy, w, h); the network predicts the
    centre shift in units of variance[0] * prior size and the log of the size ratio in units of variance[1]:
        centre = prior_centre + t_xy * v0 * prior_wh,   size = prior_wh * exp(t_wh * v1),
    returned in corner form (centre -+ size / 2)."""
    import torch
    v_centre, v_size = float(variances[0]), float(variances[1])
    p_centre, p_size = priors[:, :2], priors[:, 2:]
    centre = p_centre + loc[:, :2] * v_centre * p_size
    size = p_size * torch.exp(loc[:, 2:] * v_size)
    top_left = centre - size / 2
    return torch.cat((top_left, top_left + size), 1)      # x2 = x1 + w: the reference's own rounding of the far corner


def _iou_one_to_many(box, area_box, others, area_others):
    """IoU of one corner-form box with each row of `others`."""
    import torch
    lo = torch.maximum(others[:, :2], box[:2])
    hi = torch.minimum(others[:, 2:], box[2:])
    wh = (hi - lo).clamp(min=0.0)
    inter = wh[:, 0] * wh[:, 1]
    return inter / ((area_others - inter) + area_box)


def nms(boxes, scores, overlap: float = 0.5, top_k: int = 200):
    """Greedy non-maximum suppression (semantics of box_utils.py:175-239): among the top_k highest scores, repeatedly keep the best
    remaining box and drop every remaining box whose IoU with it exceeds `overlap` (IoU <= overlap survives).  Returns the kept
    indices, best first.  Candidates are walked in ascending-score order from the back, as the reference's sort leaves them,
    so ties resolve the same way."""
    import torch
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    order = scores.sort(0).indices[-top_k:]                # ascending; the best candidate is the last element
    kept: List[int] = []
    while order.numel() > 0:
        best = order[-1]
        kept.append(int(best))
        order = order[:-1]
        if order.numel() == 0:
            break
        iou = _iou_one_to_many(boxes[best], area[best], boxes[order], area[order])
        order = order[iou <= overlap]
    return torch.tensor(kept, dtype=torch.long, device=boxes.device)


def l2norm(x, weight, eps: float = 1e-10):
    """l2norm.py:19-24: x / (||x||_2 over channels + eps) * weight[c]."""
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + eps
    return weight.view(1, -1, 1, 1) * (x / norm)


def detect(loc, conf, priors, num_classes: int, top_k: int = 200, conf_thresh: float = 0.01, nms_thresh: float = 0.45,
           variance: Sequence[float] = (0.1, 0.2)):
    """Per-image, per-class detection list (semantics of detection.py:27-62).  loc [B, P, 4] offsets, conf [B, P, classes]
    class probabilities -> [B, classes, top_k, 5] rows (score, x1, y1, x2, y2); class 0 is background and stays empty; unused
    rows are zero.  For every foreground class: candidates above conf_thresh, decoded boxes, NMS, best first."""
    import torch
    n_img = loc.size(0)
    result = torch.zeros(n_img, num_classes, top_k, 5, device=loc.device)
    for b in range(n_img):
        boxes_b = decode(loc[b], priors, variance)
        for cls in range(1, num_classes):
            p = conf[b, :, cls]
            sel = p > conf_thresh
            if not bool(sel.any()):
                continue
            cand_scores, cand_boxes = p[sel], boxes_b[sel]
            keep = nms(cand_boxes, cand_scores, nms_thresh, top_k)
            result[b, cls, :keep.numel(), 0] = cand_scores[keep]
            result[b, cls, :keep.numel(), 1:] = cand_boxes[keep]
    return result


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
