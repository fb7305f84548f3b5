"""TEST INFRASTRUCTURE -- whole-network CPU reference built from the oracle primitives.

Walks the same k* tables as the reference host (main.cpp:36-54 -> NetWork::Init ->
Runner::Run): Quantization -> LoadModel -> (per image) LoadInputImage/feature_trans ->
input quantisation -> every layer (conv, BN requant, ReLU, pool, residual, global average)
-> Evaluation.  Returns every layer's output so the GPU path can be checked layer by
layer.  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional

import numpy as np

from . import oracle as O
from tf2_amd import config as cfg      # table/plan logic only (no device code involved)


class RefNet:
    def __init__(self, tables: cfg.NetTables, q_vals, model: np.ndarray):
        self.tables = tables
        self.plan = cfg.build_plan(tables)
        n_concat = max([L.concat for L in self.plan] + [-(L.src + 2) for L in self.plan if L.src <= -2] + [-1]) + 1
        self.n_concat = n_concat
        self.q, used = O.q_table(np.asarray(q_vals, np.int32), tables)
        self.q_used = used
        self.codes: List[Optional[np.ndarray]] = []
        self.bn: List[tuple] = []
        model = np.ascontiguousarray(model, np.float32).ravel()
        pos = 0
        for L in self.plan:
            q_in = self.q[L.q_in_row]
            q_out = self.q[L.index + 1]
            if not L.ipool:
                cnt = L.N * L.model_C * L.model_k * L.model_k
                w = model[pos:pos + cnt].reshape(L.N, L.model_C, L.model_k, L.model_k); pos += cnt
                codes = O.encode_filters(w, q_in, q_out)
            elif L.ipool == 2:                   # L2Norm row: N float scale weights
                codes = model[pos:pos + L.N].copy(); pos += L.N
            else:
                codes = None
            bias = None
            if L.bias_en:
                bias = model[pos:pos + L.N]; pos += L.N
            bn = None
            if L.bn_en:
                mean = model[pos:pos + L.N]; pos += L.N
                var = model[pos:pos + L.N]; pos += L.N
                sf = float(model[pos]); pos += 1
                gamma = model[pos:pos + L.N]; pos += L.N
                beta = model[pos:pos + L.N]; pos += L.N
                bn = (mean, var, sf, gamma, beta)
            self.bn.append(O.fold_bias_bn(L.N, q_out, bias, bn))
            self.codes.append(codes)
        assert pos == model.size, (pos, model.size)
        if tables.get("xConv1Rewrite", 0):
            self.codes[0] = O.conv1_rewrite(self.codes[0])

    def prepare_input(self, images: np.ndarray) -> np.ndarray:
        """float32 [B,C,H,W] (or int8 already quantised) -> int8 network input."""
        rewrite = self.tables.get("xConv1Rewrite", 0)
        q0 = int(self.q[0, 0])
        out = []
        for img in images:
            if img.dtype == np.int8:
                if rewrite:
                    # the transform is a pure index shuffle: apply it to the quantised values
                    f = O.feature_trans(img.astype(np.float32))
                    out.append(f.astype(np.int8))
                else:
                    out.append(img)
            else:
                f = O.feature_trans(img) if rewrite else img
                out.append(O.quantize_input(f, q0))
        return np.stack(out)

    def run(self, images: np.ndarray, upto: Optional[int] = None, times: Optional[list] = None) -> Dict[int, np.ndarray]:
        x0 = self.prepare_input(images)
        outs: Dict[int, np.ndarray] = {-1: x0}
        concat: Dict[int, np.ndarray] = {}
        B = x0.shape[0]
        for L in self.plan:
            if upto is not None and L.index > upto:
                break
            t0 = time.perf_counter()
            x = outs[L.src] if L.src >= -1 else concat[-(L.src + 2)]
            if L.ipool == 2:
                y = np.stack([O.l2norm(xi, self.q[L.q_in_row], self.q[L.index + 1], self.codes[L.index]) for xi in x])
            elif L.ipool:
                y = np.stack([O.maxpool(xi, L.pool_S, L.pool_st, L.pool_pad, L.PH, L.PW) for xi in x])
            else:
                res = outs[L.add_src] if L.add_src >= 0 else None
                b, a, be = self.bn[L.index]
                y = O.layer(L.oracle_spec(), x, self.codes[L.index], b, a, be, res)
            outs[L.index] = y
            if L.concat >= 0:
                if L.concat not in concat:
                    ctot = max(M.n_start + M.N for M in self.plan if M.concat == L.concat)
                    concat[L.concat] = np.zeros((B, ctot) + y.shape[2:], np.int8)
                concat[L.concat][:, L.n_start:L.n_start + L.N] = y
            if times is not None:
                times.append(time.perf_counter() - t0)
        return outs

    def logits(self, outs) -> np.ndarray:
        last = outs[len(self.plan) - 1]
        return last.reshape(last.shape[0], -1)

    def top5(self, logits_row):
        return O.topk(logits_row, self.q[len(self.plan)][:logits_row.size], 5)
