"""4-bit packed model file (TransForm_Kit/Compression/compress_net/4bit_data_format.txt): writer, the library's
reader (C ABI) and the independent numpy reader agree; a model loaded from it is bit-identical to the float32 load."""
import ctypes as C
import struct

import numpy as np
import pytest

from tf2_amd import _lib, config as cfg, model4bit, network, synth


def c_decode(data: bytes) -> np.ndarray:
    buf = np.frombuffer(data, np.uint8)
    n = C.c_size_t(0)
    _lib.check(_lib.lib().tf2_model4bit_decode(buf.ctypes.data, buf.size, None, 0, C.byref(n)))
    out = np.empty(n.value, np.float32)
    _lib.check(_lib.lib().tf2_model4bit_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.size, C.byref(n)))
    return out


def test_codebook_matches_the_documented_table():
    # 4bit_data_format.txt:17-37: min_exp = -6: codes 0..6 = -2^-6 .. -2^0, 7 = 0, 8..14 = +2^-6 .. +2^0
    w = np.array([-0.015625, -0.03125, -0.0625, -0.125, -0.25, -0.5, -1.0, 0.0,
                  0.015625, 0.03125, 0.0625, 0.125, 0.25, 0.5, 1.0, 0.0], np.float32).reshape(1, 16, 1, 1)
    data = model4bit.encode_tensor(w, True)
    assert struct.unpack_from("<bbhhhh", data, 0) == (-6, 0, 1, 16, 1, 1)
    words = np.frombuffer(data, "<u2", 4, 10)
    codes = np.stack([(words >> (4 * j)) & 15 for j in range(4)], 1).ravel()
    assert codes.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 7]
    np.testing.assert_array_equal(c_decode(data), w.ravel())


@pytest.mark.parametrize("shape,words", [((2, 3, 1, 1), 2), ((2, 2, 2, 2), 8), ((2, 2, 3, 3), 12),
                                          ((1, 2, 5, 5), 20), ((1, 3, 7, 7), 63), ((3, 1, 1, 4), 6)])
def test_word_grouping_per_kernel_size(shape, words):
    # 4bit_data_format.txt:38-44: 1x1 -> 4 per word, 2x2 -> 2, 3x3 -> 3, nxn -> floor(n/3) words of 3 + one remainder word
    rng = np.random.default_rng(sum(shape))
    w = (np.ldexp(1.0, -rng.integers(1, 8, size=shape)) * rng.choice([-1.0, 1.0], size=shape)).astype(np.float32)
    w[rng.random(shape) < 0.2] = 0
    data = model4bit.encode_tensor(w, True)
    assert len(data) == 10 + 2 * words
    np.testing.assert_array_equal(c_decode(data), w.ravel())
    np.testing.assert_array_equal(model4bit.read_model_4bit(data), w.ravel())


def test_unrepresentable_filters_stay_float_and_bad_input_is_rejected():
    w = np.array([0.3, -0.5, 0.25, 1.0], np.float32).reshape(1, 4, 1, 1)           # 0.3 is no power of two
    data = model4bit.encode_tensor(w, True)
    assert struct.unpack_from("<bb", data, 0)[1] == 1 and len(data) == 10 + 16
    np.testing.assert_array_equal(c_decode(data), w.ravel())
    wide = np.array([2.0 ** -9, 1.0], np.float32).reshape(1, 2, 1, 1)                # 10 exponents apart
    assert struct.unpack_from("<bb", model4bit.encode_tensor(wide, True), 0)[1] == 1
    good = model4bit.encode_tensor(np.full((1, 4, 1, 1), 0.5, np.float32), True)
    for bad in (good[:7], good[:11], good[:10] + b"\xff\xff", struct.pack("<bbhhhh", 0, 2, 1, 1, 1, 1),
                struct.pack("<bbhhhh", 0, 0, 0, 1, 1, 1)):
        buf = np.frombuffer(bad, np.uint8)
        rc = _lib.lib().tf2_model4bit_decode(buf.ctypes.data, buf.size, None, 0, None)
        assert rc != 0 and b"4-bit model" in _lib.lib().tf2_last_error()
        with pytest.raises(ValueError):
            model4bit.read_model_4bit(bad)


def test_resnet50_roundtrip_and_identical_load():
    t = cfg.resnet50_tables()
    import os
    qv = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, qv, 3)
    data = model4bit.write_model_4bit(t, model)
    # 25.5 M weights at 4 bits (3 per 16-bit word for 3x3 and 7x7 rows) + float BN/bias tensors: ~3x smaller than f32/2
    assert len(data) < 0.16 * model.nbytes
    dec = c_decode(data)
    np.testing.assert_array_equal(dec, model)
    np.testing.assert_array_equal(model4bit.read_model_4bit(data), model)
    a = network.NetWork(t); a.Quantization(synth.q_text(qv)); a.LoadModel(model)
    b = network.NetWork(t); b.Quantization(synth.q_text(qv)); b.LoadModel4bit(data)
    for l in (0, 1, 3, 26, 52, 53):
        np.testing.assert_array_equal(a.codes(l), b.codes(l))
        for x, y in zip(a.bias_bn(l), b.bias_bn(l)):
            np.testing.assert_array_equal(x, y)
