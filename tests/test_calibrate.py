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
def test_calibrator_on_gpu_squeezenet():
    import torch
    t = cfg.squeezenet11_tables(image_hw=99)
    q0 = synth.synth_q_values(t, 3, spread=1)
    model = synth.synth_model(t, q0, 3)
    imgs = synth.synth_images(t, 8, 4)
    a = calibrate.Calibrator(t, model, device="cuda:0"); a.observe(imgs)
    b = calibrate.Calibrator(t, model, device="cpu"); b.observe(imgs)
    ra, rb = a.q_rows(), b.q_rows()
    agree = sum(int((ra[k] == rb[k]).sum()) for k in ra); total = sum(ra[k].size for k in ra)
    assert agree >= 0.98 * total                      # float32 conv differences may flip a rounding boundary
    net = network.NetWork(t)
    net.Quantization(a.q_file_text().encode())
    assert net.q_values_read == cfg.q_value_count(t)
