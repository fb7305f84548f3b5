"""HIP streams created with hipExtStreamCreateWithCUMask.

Rounds 2-3 built "XCD partitions" with this: every in-flight batch on its own pair of XCDs, mask bits chosen as (i % 8) in a set of
XCD residues.  Round 4 measured what such a stream really uses (tools/ubench/cumask_probe.hip, profiles/r04_ubench_cumask_probe.txt:
every block records its XCC id and hardware id): the INTERLEAVED masks are ignored -- a stream masked to the bits i % 8 in {0, 1}
(or i % 8 == 0, or i % 4 == 0) runs on all 256 CUs of all 8 XCDs at a plain stream's speed -- and only CONTIGUOUS bit ranges restrict:
[0, 64) gives 64 CUs, eight on EVERY XCD.  So partitioned_streams() below never partitioned anything (its +2...7 % in round 3 was run-to-run
noise), real CU partitions lose badly (round 3, experiment 30), and bench.py uses plain streams now.  The functions stay for the
tests and tools that create masked streams (the library asks a stream for its CU count before it selects group launches:
Net::run, conv_bgroup.hip)."""
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


def masked_stream(bits, device="cuda:0"):
    """One torch stream whose CU mask has exactly `bits` set (contiguous ranges are what the hardware honours: [0, 32) = 32 CUs,
    four on every XCD)."""
    import torch
    dev = torch.device(device)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (n_cu + 31) // 32
    arr = (C.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    with torch.cuda.device(dev):
        rc = _hiplib().hipExtStreamCreateWithCUMask(C.byref(h), words, arr)
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(h.value, device=dev)


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
