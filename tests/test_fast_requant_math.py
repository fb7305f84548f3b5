"""The range proof behind the 3-instruction requantisation (tf2_amd/csrc/weight_pack.cpp, requant_epilogue.h):
whenever its three conditions hold, y = (acc * (alpha << lo) + bias * alpha + (beta << 20) + 2^34) >> 35 equals the
reference chain t = (int32)((int64)v * alpha >> 20); y = (((t + beta) >> 14) + 1) >> 1 with v = bias + (acc << lo)
(int32 wrap-around everywhere, pe.cl:185-203) -- checked with Python integers at the extremes of the proven range."""
import numpy as np


def wrap32(x):
    return (x + 2 ** 31) % 2 ** 32 - 2 ** 31


def reference(acc, lo, bias, alpha, beta):
    v = wrap32(bias + wrap32(acc << lo))
    t = wrap32((v * alpha) >> 20)
    return ((wrap32(t + beta) >> 14) + 1) >> 1


def generic_kernel(acc, lo, bias, alpha, beta):
    """The 6-instruction form every layer may use: beta << 20 folded into the 64-bit multiply-add, saturating +2^14."""
    v = wrap32(bias + wrap32(acc << lo))
    x = wrap32((v * alpha + (beta << 20)) >> 20)
    return min(x + 2 ** 14, 2 ** 31 - 1) >> 15


def fast_kernel(acc, lo, bias, alpha, beta):
    return (acc * wrap32(alpha << lo) + bias * alpha + (beta << 20) + 2 ** 34) >> 35


def provable(amax, lo, bias, alpha, beta):
    return (amax + abs(bias) < 2 ** 31 and (abs(alpha) << lo) < 2 ** 31 and
            (amax + abs(bias)) * abs(alpha) + (abs(beta) << 20) + 2 ** 34 < 2 ** 51)


def clamp8(y):
    return max(-128, min(127, y))


def test_fast_form_equals_reference_inside_the_proven_range():
    rng = np.random.default_rng(17)
    checked = 0
    for _ in range(4000):
        lo = int(rng.integers(0, 20))
        amax_acc = int(2 ** rng.uniform(4, 30 - lo))            # bound of |acc << lo| = amax
        amax = amax_acc << lo
        bias = int(rng.integers(-2 ** 30, 2 ** 30) >> int(rng.integers(0, 24)))
        alpha = int(rng.integers(-2 ** 24, 2 ** 24) >> int(rng.integers(0, 16)))
        beta = int(rng.integers(-2 ** 30, 2 ** 30) >> int(rng.integers(0, 12)))
        if not provable(amax, lo, bias, alpha, beta):
            continue
        for acc in (amax_acc, -amax_acc, 0, 1, -1, int(rng.integers(-amax_acc, amax_acc + 1))):
            want = reference(acc, lo, bias, alpha, beta)
            assert fast_kernel(acc, lo, bias, alpha, beta) == want
            assert clamp8(generic_kernel(acc, lo, bias, alpha, beta)) == clamp8(want)
            checked += 1
    assert checked > 3000


def test_generic_form_equals_reference_under_wrap_around():
    """Outside the proven range only the generic form is used; it must follow the reference through int32 wrap-around
    (the two differ before the clamp only where both exceed 127: x + 2^14 saturating vs the reference's 2^16)."""
    rng = np.random.default_rng(18)
    for _ in range(20000):
        lo = int(rng.integers(0, 31))
        acc = int(rng.integers(-2 ** 31, 2 ** 31))
        bias = int(rng.integers(-2 ** 31, 2 ** 31))
        alpha = int(rng.integers(-2 ** 31, 2 ** 31))
        beta = int(rng.integers(-2 ** 31, 2 ** 31))
        assert clamp8(generic_kernel(acc, lo, bias, alpha, beta)) == clamp8(reference(acc, lo, bias, alpha, beta))


def semi_kernel(acc, lo, bias, alpha, beta):
    """The 4-instruction form for rows whose v may wrap (requant_epilogue.h SEMI): the wrap of v kept, then one 64-bit
    multiply-add and one shift."""
    v = wrap32(bias + wrap32(acc << lo))
    return (v * alpha + (beta << 20) + 2 ** 34) >> 35


def semi_provable(alpha, beta):
    return (abs(alpha) << 31) + (abs(beta) << 20) + 2 ** 34 < 2 ** 51


def test_semi_form_equals_reference_when_v_wraps_but_x_cannot():
    """weight_pack.cpp packs a layer SEMI when 2^31 |alpha| + |beta << 20| + 2^34 < 2^51 for every row: then the form equals
    the reference for EVERY 32-bit accumulator, bias and shift, wrapped or not (before and after the clamp)."""
    rng = np.random.default_rng(19)
    checked = 0
    for _ in range(40000):
        lo = int(rng.integers(0, 31))
        acc = int(rng.integers(-2 ** 31, 2 ** 31))
        bias = int(rng.integers(-2 ** 31, 2 ** 31))
        alpha = int(rng.integers(-2 ** 20, 2 ** 20) >> int(rng.integers(0, 12)))
        beta = int(rng.integers(-2 ** 30, 2 ** 30) >> int(rng.integers(0, 12)))
        if not semi_provable(alpha, beta):
            continue
        want = reference(acc, lo, bias, alpha, beta)
        got = semi_kernel(acc, lo, bias, alpha, beta)
        assert got == want, (acc, lo, bias, alpha, beta)
        checked += 1
    assert checked > 30000
    # at the edge of the criterion: alpha just below 2^20 - something, extreme v
    for alpha in (2 ** 19, -(2 ** 19), 2 ** 20 - 2 ** 14, -(2 ** 20 - 2 ** 14)):
        for beta in (0, 2 ** 29, -(2 ** 29)):
            if not semi_provable(alpha, beta):
                continue
            for v in (2 ** 31 - 1, -(2 ** 31), 0, 1, -1):
                assert semi_kernel(v, 0, 0, alpha, beta) == reference(v, 0, 0, alpha, beta)
