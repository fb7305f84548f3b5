"""SSD300: the float post-processing against the reference's own functions
(tests/golden/ref_ssd.npz), and the integer table program (dilated conv6, ceil-mode / stride-1 pools, independent
pooling row, twelve head convs on six source maps) through oracle and packed-image emulation on the CPU; the GPU test
runs it on the kernels."""
import os
import sys

import numpy as np
import pytest

from tf2_amd import config as cfg, ssd, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x))


def test_prior_boxes_decode_nms_l2norm_match_reference(golden_dir):
    G = np.load(os.path.join(golden_dir, "ref_ssd.npz"))
    pri = ssd.prior_boxes(ssd.VOC)
    assert pri.shape == (8732, 4)
    np.testing.assert_array_equal(pri.numpy(), G["priors_voc"])
    # exp() may differ in the last bit between hosts
    np.testing.assert_allclose(ssd.decode(_t(G["loc"]), pri, ssd.VOC["variance"]).numpy(), G["decoded"], rtol=2e-6, atol=1e-7)
    for i in range(4):
        thr, topk = G[f"nms{i}_par"]
        keep = ssd.nms(_t(G[f"nms{i}_boxes"]), _t(G[f"nms{i}_scores"]), float(thr), int(topk))
        np.testing.assert_array_equal(keep.numpy(), G[f"nms{i}_keep"])
    np.testing.assert_allclose(ssd.l2norm(_t(G["l2_x"]), _t(G["l2_w"])).numpy(), G["l2_y"], rtol=2e-6, atol=1e-7)


def test_ssd300_table_program_shapes():
    t = cfg.ssd300_tables()
    plan = cfg.build_plan(t)
    assert len(plan) == 37 and [L.ipool for L in plan if L.ipool] == [2, 1]       # conv4_3's L2Norm row, pool4's pooling row
    conv6 = [L for L in plan if L.dil == 6][0]
    assert (conv6.k, conv6.pad_h, conv6.H, conv6.OH, conv6.N) == (3, 6, 19, 19, 1024)
    maps = [plan[l].OH for l, _ in ssd.head_rows(plan)]
    assert maps == ssd.VOC["feature_maps"]
    nbox = sum(plan[l].OH * plan[l].OW * plan[l].N // 4 for l, _ in ssd.head_rows(plan))
    assert nbox == 8732
    assert plan[6].PH == 38                                                 # ceil-mode pool3: 75 -> 38
    pool5 = [L for L in plan if L.pool_en and L.pool_st == 1]
    assert len(pool5) == 1 and pool5[0].pool_S == 3 and pool5[0].pool_pad == 1 and pool5[0].PH == 19
    # conv4_3 (row 9) feeds its heads before pool4 and through its L2Norm row (SSD.py:46-47)
    assert plan[10].ipool == 2 and plan[10].src == 9 and plan[11].ipool == 1 and plan[11].src == 9
    assert plan[25].src == 10 and plan[26].src == 10 and plan[25].H == 38 and plan[12].src == 11


def test_ssd300_small_width_oracle_vs_packed_emulation():
    import emu_packed as emu
    from oracle import netref
    from tf2_amd import network
    t = cfg.ssd300_tables(width_div=16)
    q = synth.synth_q_values(t, 3, spread=1)
    model = synth.synth_model(t, q, 3)
    ref = netref.RefNet(t, q, model)
    x = synth.synth_images(t, 1, 5)
    outs = ref.run(x)
    net = network.NetWork(t); net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    blob = net.packed_host()
    _, pls = emu.parse(blob)
    plan = cfg.build_plan(t)
    for l in (15, 18, 20, 25, 28):                    # conv6 (dilated), stride-2 extras, head rows (25: on the L2Norm row)
        L = plan[l]
        xin = outs[L.src] if L.src >= 0 else x
        got = emu.conv_from_packed(blob, pls[l], L, emu.nhwc(xin, int(pls[l]["Cp_in"])))
        np.testing.assert_array_equal(got, outs[l], err_msg=f"layer {l}")


@pytest.mark.gpu
def test_ssd300_on_gpu_every_layer_and_detect():
    """Quarter-width SSD300 at full resolution on the GPU: every row against the oracle (dilated conv6, ceil-mode and
    stride-1 pools, the independent pooling row, the twelve heads), then heads -> loc/conf -> softmax -> Detect."""
    import torch
    from test_gpu_parity import Rig
    t = cfg.ssd300_tables(width_div=4)
    q = synth.synth_q_values(t, 3, spread=1)
    model = synth.synth_model(t, q, 3)
    rig = Rig(t, q, model, 0)
    x = synth.synth_images(t, 2, 5)
    rig.check_all_layers(x)
    plan = rig.ref.plan
    qrows, pos = {}, 3
    for L in plan:
        if not L.ipool:
            qrows[L.index] = q[pos:pos + L.N]; pos += L.N
        elif L.ipool == 2:
            pos += L.N
    loc, conf = ssd.gather_heads(lambda l: rig.runner.read_layer(l, 2), plan, qrows, 2, 21)
    assert loc.shape == (2, 8732, 4) and conf.shape == (2, 8732, 21)
    out = ssd.detect(loc * 0.05, torch.softmax(conf, -1), ssd.prior_boxes(), 21, top_k=20, conf_thresh=0.2)
    assert out.shape == (2, 21, 20, 5) and torch.isfinite(out).all()


def test_l2norm_row_integer_form_tracks_the_float_op(golden_dir):
    """The L2Norm row's integer definition (oracle/tf2_oracle.c tf2o_l2norm: dequantise, IEEE double in a fixed order,
    requantise half away from zero) against the reference's float module (l2norm.py:19-24, pinned by ref_ssd.npz): the
    dequantised result is within half an output LSB of the float op applied to the dequantised input."""
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    C, H = 48, 6
    x = rng.integers(0, 128, size=(C, H, H)).astype(np.int8)
    x[:, 0, 0] = 0                                                 # an all-zero pixel: 0 / (0 + 1e-10) = 0
    Qx = rng.integers(1, 4, C); Qy = rng.integers(1, 3, C)
    w = rng.uniform(12, 24, C).astype(np.float32); w[5] = -7.5
    y = O.l2norm(x, (-Qx).astype(np.int8), (-Qy).astype(np.int8), w)
    xf = _t(x.astype(np.float32) / np.exp2(Qx.astype(np.float32))[:, None, None])[None]
    want = ssd.l2norm(xf, _t(w))[0].numpy() * np.exp2(Qy.astype(np.float32))[:, None, None]
    assert np.abs(y.astype(np.float32) - np.clip(want, -128, 127)).max() <= 0.5 + 1e-3
    assert (y[:, 0, 0] == 0).all() and (y[5] <= 0).all() and np.abs(y).max() > 20
    # the formula the module is pinned by
    G = np.load(os.path.join(golden_dir, "ref_ssd.npz"))
    np.testing.assert_allclose(ssd.l2norm(_t(G["l2_x"]), _t(G["l2_w"])).numpy(), G["l2_y"], rtol=2e-6, atol=1e-7)
