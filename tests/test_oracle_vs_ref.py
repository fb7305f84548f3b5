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
