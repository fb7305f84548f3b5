"""A HIP stream created with hipExtStreamCreateWithCUMask, for the tests and tools that need a CU-restricted stream (the library asks a
stream for its CU count before it selects group launches one batch at a time: Net::run, conv_bgroup.hip).

Only CONTIGUOUS mask bit ranges restrict a stream on gfx950 ([0, 64) = 64 CUs, eight on every XCD); interleaved masks are ignored by the
hardware (tools/ubench/cumask_probe.hip, profiles/r04_ubench_cumask_probe.txt) -- which is why rounds 2-3's "XCD-partitioned streams"
were plain streams under another name; they left the tree in round 5, and bench.py runs its batches in flight on plain streams.  The
library's 64-CU test for group launches counts MASK BITS (hipExtStreamGetCUMask), not granted CUs: an interleaved 32-bit mask is
treated as a small stream although it runs on the whole chip -- safe (no group launches), and noted in include/tf2_amd.h."""
from __future__ import annotations

import ctypes as C

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
