"""TransForm_Kit's 4-bit packed model file: writer + independent reader.

Format: `TransForm_Kit/Compression/compress_net/4bit_data_format.txt:1-44` -- the reference documents it and
ships no code.  The canonical byte layout is spelled out in `tf2_amd/csrc/model4bit.cpp` (the product's reader,
behind `tf2_model4bit_decode` / `tf2_net_load_model_4bit`); this module is the WRITER (the TransForm_Kit side,
which is Python in the reference) and a numpy reader used to cross-check the C++ one.

A model is the sequence of parameter tensors in `LoadModel` order (`Runtime_Engine/cnn/host/src/
model_loader.cpp:154-213`): per conv/FC layer the filter `[N][C][k][k]` (4-bit codes when every weight is 0 or
+-2^e with e inside a 7-exponent window, else float32), then `[bias]`, then `[mean, var, scale_factor, gamma,
beta]` as float32 tensors.
"""
from __future__ import annotations

import struct
from typing import List, Tuple

import numpy as np

from . import config as cfg


def split_model(tables: cfg.NetTables, model: np.ndarray) -> List[Tuple[str, np.ndarray, bool]]:
    """float32 LoadModel stream -> [(name, 4-D array, is_filter)] in file order."""
    model = np.ascontiguousarray(model, np.float32).ravel()
    out, pos = [], 0

    def take(name, shape, is_filter=False):
        nonlocal pos
        n = int(np.prod(shape))
        out.append((name, model[pos:pos + n].reshape(shape), is_filter))
        pos += n

    for L in cfg.build_plan(tables):
        if not L.ipool:
            take(f"layer{L.index}.filter", (L.N, L.model_C, L.model_k, L.model_k), True)
        elif L.ipool == 2:
            take(f"layer{L.index}.l2w", (L.N, 1, 1, 1))
        if L.bias_en:
            take(f"layer{L.index}.bias", (L.N, 1, 1, 1))
        if L.bn_en:
            for nm, n in (("mean", L.N), ("var", L.N), ("scale_factor", 1), ("gamma", L.N), ("beta", L.N)):
                take(f"layer{L.index}.{nm}", (n, 1, 1, 1))
    if pos != model.size:
        raise ValueError(f"model stream has {model.size} floats, the tables need {pos}")
    return out


def _codes_for(w: np.ndarray):
    """(min_exp, codes uint8) if every weight is 0 or +-2^e within one 7-exponent window, else None."""
    flat = w.ravel()
    nz = flat != 0
    if not nz.any():
        return -6, np.full(flat.shape, 7, np.uint8)
    m, e = np.frexp(np.abs(flat[nz]))             # |w| = m * 2^e, m in [0.5, 1)
    if not np.all(m == 0.5):
        return None
    ex = e.astype(np.int64) - 1
    lo = int(ex.min())
    if int(ex.max()) - lo > 6 or not (-40 <= lo <= 20):
        return None
    codes = np.full(flat.shape, 7, np.uint8)
    k = (ex - lo).astype(np.uint8)
    codes[nz] = np.where(flat[nz] < 0, k, k + 8)
    return lo, codes


def _pack_words(codes: np.ndarray, N: int, C: int, H: int, W: int) -> np.ndarray:
    rows = N * C * H
    c = codes.reshape(rows, W).astype(np.uint16)
    if W == 1:
        flat = np.zeros(((rows + 3) // 4) * 4, np.uint16)
        flat[:rows] = c[:, 0]
        q = flat.reshape(-1, 4)
        return (q[:, 0] | (q[:, 1] << 4) | (q[:, 2] << 8) | (q[:, 3] << 12)).astype("<u2")
    wpr = W // 3 + (1 if W % 3 else 0)
    padded = np.zeros((rows, wpr * 3), np.uint16)
    padded[:, :W] = c
    g = padded.reshape(rows, wpr, 3)
    return (g[..., 0] | (g[..., 1] << 4) | (g[..., 2] << 8)).astype("<u2").ravel()


def encode_tensor(t: np.ndarray, is_filter: bool) -> bytes:
    t = np.ascontiguousarray(t, np.float32)
    N, C, H, W = t.shape
    if max(t.shape) > 32767:
        raise ValueError("dimension does not fit the int16 header field")
    enc = _codes_for(t) if is_filter else None
    if enc is None:
        return struct.pack("<bbhhhh", 0, 1, N, C, H, W) + t.tobytes()
    min_exp, codes = enc
    return struct.pack("<bbhhhh", min_exp, 0, N, C, H, W) + _pack_words(codes, N, C, H, W).tobytes()


def write_model_4bit(tables: cfg.NetTables, model: np.ndarray) -> bytes:
    """float32 LoadModel stream -> bytes of the 4-bit packed model file."""
    return b"".join(encode_tensor(t, f) for _, t, f in split_model(tables, model))


def read_model_4bit(data: bytes) -> np.ndarray:
    """Independent numpy reader: 4-bit packed model bytes -> float32 LoadModel stream."""
    out, pos, n = [], 0, len(data)
    while pos < n:
        if n - pos < 10:
            raise ValueError("truncated tensor header")
        min_exp, dtype, N, C, H, W = struct.unpack_from("<bbhhhh", data, pos)
        pos += 10
        if min(N, C, H, W) <= 0:
            raise ValueError("non-positive dimension")
        cnt = N * C * H * W
        if dtype == 1:
            out.append(np.frombuffer(data, "<f4", cnt, pos)); pos += 4 * cnt
        elif dtype == 0:
            rows = N * C * H
            if W == 1:
                words = np.frombuffer(data, "<u2", (rows + 3) // 4, pos); pos += 2 * words.size
                codes = np.stack([(words >> (4 * j)) & 15 for j in range(4)], 1).ravel()[:rows]
            else:
                wpr = W // 3 + (1 if W % 3 else 0)
                words = np.frombuffer(data, "<u2", rows * wpr, pos).reshape(rows, wpr); pos += 2 * words.size
                codes = np.stack([(words >> (4 * j)) & 15 for j in range(3)], 2).reshape(rows, wpr * 3)[:, :W].ravel()
            if (codes == 15).any():
                raise ValueError("unused code 15")
            k = codes.astype(np.int64)
            mag = np.ldexp(1.0, min_exp + np.where(k < 7, k, k - 8)).astype(np.float32)
            out.append(np.where(k == 7, np.float32(0), np.where(k < 7, -mag, mag)).astype(np.float32))
        else:
            raise ValueError(f"unknown data type {dtype}")
    return np.concatenate(out) if out else np.zeros(0, np.float32)
