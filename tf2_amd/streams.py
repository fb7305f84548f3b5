"""HIP streams restricted to whole XCDs (hipExtStreamCreateWithCUMask).

MI355X has 8 XCDs of 32 CUs, each with its own L2.  A server that keeps N independent batches in flight can give every
batch its own set of XCDs: the kernels of one batch then never wait for CU slots behind another batch's kernels, and a batch's
activations stay in its XCDs' L2.  On this part bit i of the CU mask is CU i / 8 of XCD i % 8 (measured: masks built that way
and masks of consecutive-bit groups of 8 XCD-interleaved CUs behave alike, contiguous quarters of the bit range do not), so a
partition of k XCDs is the bits with (i % 8) in a set of k residues.  bench.py uses four partitions of two XCDs by default
(+7 % at 20 timed steps, +2 % at 100 against plain streams, profiles/r03_cumask.txt)."""
from __future__ import annotations

import ctypes as C
from typing import List

_hip = None


def _hiplib():
    global _hip
    if _hip is None:
        err = None
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError as e:             # pragma: no cover
                err = e
        if _hip is None:
            raise OSError(f"libamdhip64 not loadable: {err}")
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    return _hip


def xcd_mask_bits(part: int, n_parts: int, n_cu: int, n_xcd: int = 8) -> List[int]:
    """CU-mask bit indices of partition `part` of `n_parts` (n_parts must divide n_xcd): whole XCDs."""
    if n_xcd % n_parts:
        raise ValueError(f"{n_parts} partitions do not divide {n_xcd} XCDs")
    per = n_xcd // n_parts
    return [i for i in range(n_cu) if (i % n_xcd) // per == part]


def partitioned_streams(n_parts: int, device="cuda:0"):
    """n_parts torch streams, stream k restricted to XCDs [k * 8 / n_parts, (k + 1) * 8 / n_parts).  Raises OSError / RuntimeError
    when the runtime refuses; callers fall back to plain streams."""
    import torch
    dev = torch.device(device)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (n_cu + 31) // 32
    out = []
    with torch.cuda.device(dev):
        for k in range(n_parts):
            arr = (C.c_uint32 * words)()
            for b in xcd_mask_bits(k, n_parts, n_cu):
                arr[b // 32] |= 1 << (b % 32)
            h = C.c_void_p()
            rc = _hiplib().hipExtStreamCreateWithCUMask(C.byref(h), words, arr)
            if rc != 0 or not h.value:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
            out.append(torch.cuda.ExternalStream(h.value, device=dev))
    return out
