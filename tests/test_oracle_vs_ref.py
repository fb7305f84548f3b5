"""Live cross-check of the oracle against the reference's compiled host functions
(oracle/_ref/libtf2ref_resnet50.so, built by oracle/Makefile from /root/reference where the
sources lie).  Skipped when that build is absent."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libtf2ref_resnet50.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def R():
    try:
        L = C.CDLL(REF_SO)
    except OSError as e:           # e.g. libOpenCL missing on the box
        pytest.skip(f"cannot load reference build: {e}")
    L._Z8Get_realfc.restype = C.c_char
    L._Z8Get_realfc.argtypes = [C.c_float, C.c_char]
    return L


def test_get_real_fuzz(R):
    rng = np.random.default_rng(3)
    vals = np.concatenate([rng.normal(0, 0.3, 3000), np.ldexp(rng.choice([-1.0, 1.0], 3000) * rng.uniform(0.985, 1.015, 3000),
                                                                -rng.integers(0, 17, 3000))]).astype(np.float32)
    exps = rng.integers(-5, 40, vals.size).astype(np.int8)
    for v, e in zip(vals, exps):
        r = R._Z8Get_realfc(C.c_float(float(v)), C.c_char(int(e) & 0xff))
        assert O.get_real(v, int(e)) == r[0], (float(v), int(e))


def test_filter_trans_fuzz(R):
    rng = np.random.default_rng(4)
    for _ in range(50):
        plane = rng.integers(0, 256, 49).astype(np.uint8)
        want = np.zeros(81, np.uint8); got = np.zeros(81, np.uint8)
        R._Z12filter_transPcS_(plane.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p))
        O.lib().tf2o_filter_trans(plane.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(got, want)


def test_feature_trans_fuzz(R):
    rng = np.random.default_rng(5)
    plane = rng.normal(0, 60, 224 * 224).astype(np.float32)
    fo = np.zeros(9 * 115 * 115 + 2048, np.float32)
    R._Z13feature_transPfS_(plane.ctypes.data_as(C.c_void_p), fo.ctypes.data_as(C.c_void_p))
    want = fo[:9 * 115 * 115].reshape(9, 115, 115)[:, :114, :114]
    got = np.empty((9, 114, 114), np.float32)
    O.lib().tf2o_feature_trans(plane.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(got, want)


# ---- the device code's PE arithmetic (pe.cl:27-49), compiled in place as C (oracle/ref_pe_probe.c, oracle/Makefile `ref`) ----
PE_SO = os.path.join(os.path.dirname(REF_SO), "libtf2ref_pe.so")


@pytest.fixture(scope="module")
def PE():
    if not os.path.exists(PE_SO):
        pytest.skip("oracle/_ref/libtf2ref_pe.so not built")
    L = C.CDLL(PE_SO)
    L.tf2ref_pe_mul.restype = C.c_int
    L.tf2ref_pe_mul.argtypes = [C.c_int, C.c_int]
    L.tf2ref_pe_dot.restype = C.c_int
    L.tf2ref_pe_dot.argtypes = [C.c_void_p, C.c_void_p]
    return L


def test_pe_mul_every_feature_and_code(PE):
    """tf2o_mul == the reference's own MUL (pe.cl:27-40) on ALL 256 x 256 (feature, filter code) pairs: the zero flag (bit 6), the sign flag
    with the -128 quirk ((int8)(-x) keeps -128), five-bit shifts up to 31 with the int32 wrap of `feature << filter`."""
    for x in range(-128, 128):
        for code in range(256):
            assert O.mul(x, code) == PE.tf2ref_pe_mul(x, code), (x, hex(code))
    assert PE.tf2ref_pe_mul(-128, 0x83) == -1024 == PE.tf2ref_pe_mul(-128, 0x03) and PE.tf2ref_pe_mul(127, 0x14) == 133169152


def test_pe_dot_product_and_conv_core(PE):
    """The reference's DotProduct (pe.cl:42-49: C_VECTOR = 16 MULs summed in an int, i.e. mod 2^32) against the oracle's convolution core on
    the same 16 channels (a 1x1 layer of one output channel, bias 0): random vectors, and vectors built to overflow int32 (shift 31 / 30
    codes on +-127 / -128 features)."""
    assert PE.tf2ref_pe_c_vector() == 16
    rng = np.random.default_rng(11)
    for trial in range(3000):
        x = rng.integers(-128, 128, 16).astype(np.int8)
        if trial % 3 == 0:
            codes = (rng.integers(24, 32, 16) | (rng.integers(0, 2, 16) << 7)).astype(np.uint8)        # long shifts: the sum wraps
            x = rng.choice(np.array([-128, -127, 127], np.int8), 16)
        else:
            codes = rng.integers(0, 256, 16).astype(np.uint8)
        ref = PE.tf2ref_pe_dot(x.ctypes.data, codes.ctypes.data)
        acc = O.conv(x.reshape(16, 1, 1), codes.reshape(1, 16, 1, 1), np.zeros(1, np.int32))
        assert int(acc.reshape(-1)[0]) == ref, (trial, x.tolist(), codes.tolist())
