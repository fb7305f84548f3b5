"""CAQ calibrator (tf2_amd/calibrate.py) against the reference's own functions (tests/golden/ref_caq.npz) and
end to end: calibrated Q file -> integer engine plumbing (Quantization accepts it; the oracle run stays in range)."""
import os

import numpy as np
import pytest

from tf2_amd import calibrate, config as cfg, network, synth


def test_quantize_for_shift_matches_reference(golden_dir):
    G = np.load(os.path.join(golden_dir, "ref_caq.npz"))
    got = np.array([calibrate.quantize_for_shift(x) for x in G["fs_x"]])
    np.testing.assert_array_equal(got, G["fs_q"])
    assert calibrate.quantize_for_shift(np.zeros((2, 3))) == 0


def test_quantize_channels_matches_reference(golden_dir):
    G = np.load(os.path.join(golden_dir, "ref_caq.npz"))
    for i in range(5):
        np.testing.assert_array_equal(calibrate.quantize_channels(G[f"qc{i}_x"]), G[f"qc{i}_q"])


def test_float_forward_matches_dequantised_integer_engine_shapes_and_scale():
    """The float forward runs the same table program as the oracle: same tensor shapes for every row, and the
    integer result (oracle) dequantised with the calibrated Q tracks the float result."""
    from oracle import netref
    t = cfg.tiny_tables()
    q0 = synth.synth_q_values(t, 5, spread=0)
    model = synth.synth_model(t, q0, 5)
    imgs = synth.synth_images(t, 4, 9)
    cal = calibrate.Calibrator(t, model, device="cpu")
    cal.observe(imgs[:2]); cal.observe(imgs[2:])
    text = cal.q_file_text()
    qv = np.array([int(v) for v in text.split()], np.int32)
    assert qv.size == cfg.q_value_count(t)
    net = network.NetWork(t)
    q = net.Quantization(text.encode())
    assert net.q_values_read == qv.size
    ref = netref.RefNet(t, qv, model)
    outs_i = ref.run(imgs)
    outs_f = calibrate.float_forward(t, model, imgs, "cpu")
    plan = cfg.build_plan(t)
    rows = cal.q_rows()
    checked = 0
    for L in plan:
        f = outs_f[L.index].numpy()
        assert f.shape == outs_i[L.index].shape
        if L.ipool:
            continue
        deq = outs_i[L.index].astype(np.float64) / np.exp2(rows[L.index].astype(np.float64))[None, :, None, None]
        # calibrated Q keeps every channel inside int8 and uses most of its range somewhere
        assert np.abs(outs_i[L.index].astype(np.int32)).max() <= 128
        err = np.abs(deq - f).mean() / (np.abs(f).mean() + 1e-9)
        if L.index <= 1:
            assert err < 0.2, (L.index, err)          # early layers: quantisation noise only
        checked += 1
    assert checked >= 3


@pytest.mark.gpu
def test_calibrator_on_gpu_against_the_reference_pass(golden_dir):
    """The same comparison with the float forward on the MI355X: the Q vectors of the reference's own calibration pass
    (ref_caq_squeezenet.npz).  GPU convolutions accumulate in another order than the CPU ones the fixture was made
    with, so a channel whose max |feature| sits on a power-of-two boundary may move by one: at least 99 % identical,
    never more than 1 apart."""
    G = np.load(os.path.join(golden_dir, "ref_caq_squeezenet.npz"))
    t = cfg.squeezenet11_tables()
    stream, _ = synth.squeezenet_seeded_stream()
    cal = calibrate.Calibrator(t, stream, device="cuda:0", bn_eps=1e-3)
    for im in synth.squeezenet_calibration_images():
        cal.observe(im)
    rows = cal.q_rows()
    same = total = 0
    for L in cal.plan:
        want = G[f"row{L.index}_q"]
        assert np.abs(rows[L.index] - want).max() <= 1
        same += int((rows[L.index] == want).sum()); total += want.size
    assert same >= 0.99 * total
    net = network.NetWork(t)
    net.Quantization(cal.q_file_text().encode())
    assert net.q_values_read == cfg.q_value_count(t)


def test_squeezenet_227_reproduces_the_references_own_calibration_pass(golden_dir):
    """BASELINE configs[0] (SqueezeNet 1.1, 1x3x227x227, TransForm_Kit/Quantization CPU forward): the fixture holds the Q
    vectors the REFERENCE's calibration code (feature_write.py feature_hook + quantization.py QuantizeChannel, executed
    in the build container on the reference's own models/SqueezeNet/SqueezeNet.py) produced for seeded parameters and
    three seeded images; float_forward + Calibrator on the same table program must give the same Q for every row."""
    G = np.load(os.path.join(golden_dir, "ref_caq_squeezenet.npz"))
    t = cfg.squeezenet11_tables()
    stream, _ = synth.squeezenet_seeded_stream()
    assert stream.size == cfg.model_float_count(t)
    imgs = synth.squeezenet_calibration_images()
    cal = calibrate.Calibrator(t, stream, device="cpu", bn_eps=1e-3)          # SqueezeNet.py:23: BatchNorm2d(eps=0.001)
    for im in imgs:
        cal.observe(im)
    rows = cal.q_rows()
    np.testing.assert_array_equal(rows[-1], G["image_q"])
    n_ch = n_same = 0
    for L in cal.plan:
        want = G[f"row{L.index}_q"]
        n_ch += want.size; n_same += int((rows[L.index] == want).sum())
        np.testing.assert_array_equal(rows[L.index], want, err_msg=f"Q row of table row {L.index}")
    assert n_ch == cfg.q_value_count(t) - 3 and n_same == n_ch
    # and the float tensors themselves (max |feature| over the three images)
    np.testing.assert_allclose(cal.maxabs[0][:, :4].numpy(), G["row0_maxabs_ch0_3"], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(cal.maxabs[25].numpy(), G["row25_maxabs"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(cal.maxabs[26].numpy().reshape(1, 128, 1, 1), G["row26_maxabs"].reshape(1, 128, 1, 1), rtol=1e-3, atol=1e-3)
    # the Q file it writes is accepted by Quantization (quantization.cpp:25-55)
    net = network.NetWork(t)
    net.Quantization(cal.q_file_text().encode())
    assert net.q_values_read == cfg.q_value_count(t)
