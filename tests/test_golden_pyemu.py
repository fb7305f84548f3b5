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


# ---- the emulator's own Bottleneck.forward / ResNet.forward (…Batch-2.py:249-323, 395-443), every intermediate --------
# tests/golden/ref_pyemu_block.npz (oracle/gen_golden.py gen_pyemu_block).  The BN parameters lie on a dyadic grid
# (alpha = m / 2^12, beta = k / 2^8), so X = alpha * acc + beta * 2^(15+Q) is exact in the emulator's float arithmetic
# and in int64 here:  v = X / 2^15.  The FPGA rule (pe.cl:191-193: t = acc*alpha_fix >> 20; ((t + beta_fix) >> 14) + 1 >> 1)
# is floor(v + 1/2) exactly; the emulator rounds half to even (and stores v as float32 first), so the two differ only
# on exact ties with an even integer part -- enumerated and asserted below.

@pytest.fixture(scope="module")
def PB(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_pyemu_block.npz"))


def _fold(alpha, beta, Q):
    n = alpha.size
    q_run = (-np.asarray(Q[:n], np.int32)).astype(np.int8)
    _, af, bf = O.fold_bias_bn(n, q_run, None, (np.zeros(n, np.float32), np.ones(n, np.float32) - 1e-5, 1.0,
                                                np.asarray(alpha, np.float32), np.asarray(beta, np.float32)))
    return af, bf


def _exact_bn(acc, alpha, beta, Q):
    """(floor(v + 1/2) as int64, |frac(v) - 1/2| as float64, exact-tie mask) with v = (alpha*acc + beta*2^(15+Q)) / 2^15."""
    n = alpha.size
    m = np.round(alpha.astype(np.float64) * 4096).astype(np.int64)
    assert np.array_equal(m / 4096.0, alpha.astype(np.float64))
    bfix = np.round(beta.astype(np.float64) * 2.0 ** (15 + np.asarray(Q[:n], np.float64))).astype(np.int64)
    shp = (1, n) + (1,) * (acc.ndim - 2)
    X4096 = acc.astype(np.int64) * m.reshape(shp) + (bfix.reshape(shp) << 12)          # X * 2^12
    D = 1 << 27                                                                      # 2^15 * 2^12
    half_up = (X4096 + (D >> 1)) // D                                                # floor(v + 1/2)
    r = X4096 % D
    return half_up, np.abs(r / D - 0.5), r == (D >> 1)


def _check_requant(acc, alpha, beta, Q, emu_bn, min_safe):
    af, bf = _fold(alpha, beta, Q)
    # the fold reproduces alpha_fix = trunc(alpha * 2^20), beta_fix = round(beta * 2^(15+Q)) (model_loader.cpp:223-231)
    np.testing.assert_array_equal(af, np.round(alpha.astype(np.float64) * (1 << 20)).astype(np.int64))
    half_up, dist, tie = _exact_bn(acc, alpha, beta, Q)
    want_fpga = np.clip(half_up, -128, 127).astype(np.int8)
    ys = np.stack([O.requant(acc[b], af, bf, relu=False) for b in range(acc.shape[0])])
    np.testing.assert_array_equal(ys, want_fpga)                      # every element: the pe.cl rule, restated in int64
    emu = np.clip(emu_bn, -128, 127).astype(np.int8)
    safe = dist > 1e-3
    assert safe.sum() >= min_safe
    np.testing.assert_array_equal(ys[safe], emu[safe])                # every non-tie element: the executed emulator
    # exact ties: the FPGA rounds up, the emulator to even
    differ = tie & (ys != emu)
    np.testing.assert_array_equal(ys[differ].astype(np.int32), emu[differ].astype(np.int32) + 1)
    return int(tie.sum()), int(differ.sum()), safe


BLOCK_CONVS = {"b1": [(1, 0, 1), (1, 1, 2), (1, 0, 3)], "b2": [(1, 0, 1), (2, 1, 2), (1, 0, 3), (2, 0, 3)]}   # stride, pad, Q row


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_bottleneck_conv_sums_and_stride2(PB, tag):
    for i, (stride, pad, _) in enumerate(BLOCK_CONVS[tag]):
        codes = codes_from(PB[f"{tag}_c{i}_shift"], PB[f"{tag}_c{i}_sign"])
        x = PB[f"{tag}_conv{i}_in"]
        for b in range(x.shape[0]):
            acc = O.conv(x[b], codes, np.zeros(codes.shape[0], np.int32), stride=stride, pad=pad)
            np.testing.assert_array_equal(acc, PB[f"{tag}_conv{i}_acc"][b])


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_bottleneck_requant_clamp_relu(PB, tag):
    Q = PB[f"{tag}_q"]
    for i, (_, _, qrow) in enumerate(BLOCK_CONVS[tag]):
        acc = PB[f"{tag}_conv{i}_acc"]
        al, be = PB[f"{tag}_c{i}_alpha"], PB[f"{tag}_c{i}_beta"]
        _, _, safe = _check_requant(acc, al, be, Q[qrow], PB[f"{tag}_bn{i}_out"], int(acc.size * 0.95))
        if i < 2:
            # clamp -> ReLU -> int8 is what the next convolution consumed (relu.cl:50-56 fused in the oracle's requant)
            af, bf = _fold(al, be, Q[qrow])
            yr = np.stack([O.requant(acc[b], af, bf, relu=True) for b in range(acc.shape[0])])
            nxt = PB[f"{tag}_conv{i + 1}_in"]
            np.testing.assert_array_equal(yr[safe], nxt[safe])
            assert (nxt >= 0).all() and (yr == np.maximum(np.stack([O.requant(acc[b], af, bf, relu=False) for b in range(acc.shape[0])]), 0)).all()


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_bottleneck_residual_add_int16_clamp_relu(PB, tag):
    """feature_writer.cl:119-122 vs the emulator's `out = np.int16(out); out += identity; clamp; relu` (:300-318)."""
    main = np.clip(PB[f"{tag}_bn2_out"], -128, 127).astype(np.int8)
    ident = PB[f"{tag}_x"] if tag == "b1" else np.clip(PB[f"{tag}_bn3_out"], -128, 127).astype(np.int8)
    got = O.residual_add(main, ident, relu=True)
    np.testing.assert_array_equal(got, PB[f"{tag}_y"].astype(np.int8))
    s = main.astype(np.int32) + ident.astype(np.int32)
    assert (s > 127).any() and (s < 0).any()                  # both the clamp and the ReLU were exercised
    # without the ReLU the same sum is only clamped
    np.testing.assert_array_equal(O.residual_add(main, ident, relu=False), np.clip(s, -128, 127).astype(np.int8))


def test_head_conv1_requant_relu_maxpool(PB):
    codes = codes_from(PB["head_c0_shift"], PB["head_c0_sign"])
    acc = O.conv(PB["head_img"][0], codes, np.zeros(64, np.int32), stride=2, pad=3)
    np.testing.assert_array_equal(acc, PB["head_conv1_acc"][0])
    _check_requant(PB["head_conv1_acc"], PB["head_c0_alpha"], PB["head_c0_beta"], PB["head_q1"], PB["head_conv1_bn"], 12000)
    np.testing.assert_array_equal(np.clip(PB["head_conv1_bn"], -128, 127), PB["head_conv1_clamped"])
    relu = np.maximum(PB["head_conv1_clamped"][0], 0).astype(np.int8)
    assert (relu > 0).mean() > 0.3
    # pool.cl:178-252 (zero-extended 3-max) vs the emulator's nn.MaxPool2d(3, 2, 1) on the post-ReLU map
    np.testing.assert_array_equal(O.maxpool(relu, 3, 2, 1, 7, 7), PB["head_pool1"][0].astype(np.int8))


def _avg_rule(x):
    S = x.reshape(x.shape[0], -1).astype(np.int64).sum(1)
    S16 = ((S + 32768) % 65536) - 32768                                      # int16 accumulator, full_size_pool.cl:104-112
    want = np.clip((((S16 * 669) >> 14) + 1) >> 1, -128, 127).astype(np.int8)  # full_size_pool.cl:115-119
    frac = (S / 49.0) - np.floor(S / 49.0)
    return want, np.abs(frac - 0.5), S


def test_global_average_669_rule_vs_avgpool_round(PB):
    xs, ys = PB["avg_x"], PB["avg_y"]
    n_safe = n_all = 0
    for i in range(xs.shape[0]):
        got = O.global_avg(xs[i])
        want, dist, S = _avg_rule(xs[i])
        np.testing.assert_array_equal(got, want)                 # every element: the 669 rule restated independently
        # 669/2^15 exceeds 1/49 by 8.1e-6: the two roundings can only differ when frac(S/49) is within
        # |S| * 8.1e-6 + (half-even vs half-up) of 1/2 -- everywhere else the executed emulator must agree
        safe = dist > np.abs(S) * 8.2e-6 + 1e-6
        np.testing.assert_array_equal(got[safe], ys[i][safe].astype(np.int8))
        n_safe += int(safe.sum()); n_all += safe.size
    assert n_safe > 0.9 * n_all and n_all >= 3000
    # the map the emulator's own forward produced (post max-pool)
    got = O.global_avg(PB["head_pool1"][0].astype(np.int8))
    want, dist, S = _avg_rule(PB["head_pool1"][0].astype(np.int8))
    np.testing.assert_array_equal(got, want)
    safe = dist > np.abs(S) * 8.2e-6 + 1e-6
    np.testing.assert_array_equal(got[safe], PB["head_pool5"][0].reshape(-1)[safe].astype(np.int8))


def test_head_fc_matches_emulator(PB):
    Q = PB["head_qfc"]
    codes = codes_from(PB["head_fc_shift"], PB["head_fc_sign"]).reshape(1000, 64, 1, 1)
    x = PB["head_pool5"][0].reshape(64, 1, 1).astype(np.int8)
    # bias_fix = (int)(b * (float)(1 << (15+Q)))  (model_loader.cpp:176-188); the emulator adds b * 2^(Q+15) in float
    bias_fix = (PB["head_fc_bias"].astype(np.float64) * 2.0 ** (15 + Q.astype(np.float64))).astype(np.int64)
    acc = O.conv(x, codes, bias_fix.astype(np.int32))
    y = O.requant(acc, np.full(1000, 1 << 20, np.int32), np.zeros(1000, np.int32), relu=False).ravel()
    v = acc.ravel().astype(np.int64)
    half_up = np.clip((v + (1 << 14)) >> 15, -128, 127)
    np.testing.assert_array_equal(y, half_up.astype(np.int8))
    # the emulator holds the sums in float32 (exact below 2^24, within 2^-22 relative above): compare away from ties
    tie = np.abs((v & 0x7fff) / 32768.0 - 0.5) < 1e-3
    emu = np.clip(PB["head_fc"][0], -128, 127).astype(np.int8)
    np.testing.assert_array_equal(y[~tie], emu[~tie])
    assert (~tie).sum() > 900 and len(np.unique(emu)) > 50


def test_bn_1e5_samples_with_enumerated_ties(PB):
    n_tie, n_diff, safe = _check_requant(PB["bnx_acc"], PB["bnx_alpha"], PB["bnx_beta"], PB["bnx_q"], PB["bnx_y"], 100000)
    assert PB["bnx_acc"].size >= 100000
    assert n_tie >= 10 and n_diff >= 3          # real ties occurred, and some of them separate half-up from half-even
