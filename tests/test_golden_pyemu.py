"""Oracle device arithmetic vs the reference's Python FPGA emulator
(TransForm_Kit/Quantization/debug/...Batch-2.py: Conv2dInt8 :121-142, BN :144-158, FC :160-179),
through tests/golden/ref_pyemu.npz (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def P(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_pyemu.npz"))


def codes_from(shift, sign):
    code = shift.astype(np.uint8) & 0x1f
    code = np.where(sign < 0, code | 0x80, code)
    return np.where(sign == 0, 0x40, code).astype(np.uint8)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_conv_sums_match_conv2dint8(P, idx):
    x, shift, sign = P[f"conv{idx}_x"], P[f"conv{idx}_shift"], P[f"conv{idx}_sign"]
    stride, pad = (int(v) for v in P[f"conv{idx}_geom"])
    codes = codes_from(shift, sign)
    bias = np.zeros(codes.shape[0], np.int32)
    for b in range(x.shape[0]):
        acc = O.conv(x[b], codes, bias, stride=stride, pad=pad)
        np.testing.assert_array_equal(acc, P[f"conv{idx}_acc"][b])


def test_requant_agrees_with_float_bn_away_from_ties(P):
    acc, alpha, beta, Q = P["bn_acc"], P["bn_alpha"], P["bn_beta"], P["bn_q"]
    q_run = (-Q).astype(np.int8)
    n = alpha.size
    _, af, bf = O.fold_bias_bn(n, q_run, None, (np.zeros(n, np.float32), np.ones(n, np.float32) - 1e-5, 1.0, alpha, beta))
    # with mean 0 / var 1-eps / sf 1 the fold gives alpha_fix = trunc(alpha*2^20), beta_fix = round(beta*2^(15+Q))
    y = O.requant(acc[0], af, bf, relu=False)
    want = np.clip(P["bn_y"][0], -128, 127).astype(np.int8)
    safe = P["bn_safe"][0]
    assert safe.sum() > 500
    np.testing.assert_array_equal(y[safe], want[safe])
    # far outside the int8 range both saturate identically
    far = np.abs(P["bn_y"][0]) > 140
    np.testing.assert_array_equal(y[far], want[far])


def test_fc_matches_emulator(P):
    x, sh, sg, bias = P["fc_x"], P["fc_shift"], P["fc_sign"], P["fc_bias"]
    codes = codes_from(sh, sg).reshape(sh.shape[0], sh.shape[1], 1, 1)
    acc = O.conv(x[0].reshape(-1, 1, 1), codes, bias.astype(np.int32))
    # FC(): (sum + bias) * 2^-15, round, clamp.  The FPGA path uses alpha=2^20, beta=0:
    y = O.requant(acc, np.full(sh.shape[0], 1 << 20, np.int32), np.zeros(sh.shape[0], np.int32), relu=False).ravel()
    exact = (acc.ravel().astype(np.float64)) * 2.0 ** -15
    safe = np.abs(exact - np.floor(exact) - 0.5) > 0.02
    want = np.clip(P["fc_y"][0], -128, 127).astype(np.int8)
    np.testing.assert_array_equal(y[safe], want[safe])


def test_mul_quirk_minus128():
    # measured on the reference's pe.cl compiled as C: MUL(-128,0x83) = -1024
    assert O.mul(-128, 0x83) == -1024 == O.mul(-128, 0x03)
    assert O.mul(127, 0x14) == 133169152
    assert O.mul(5, 0x40) == 0 and O.mul(-3, 0x82) == 12 and O.mul(1, 0x1f) == -(1 << 31)
