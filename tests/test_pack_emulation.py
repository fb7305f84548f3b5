"""CPU check of the packed weight image (tf2_amd/csrc/weight_pack.cpp): a numpy model of the
kernels' data flow (tests/emu_packed.py) fed with the packed image must reproduce the oracle
layer by layer -- exponent windows + Horner recombination are exact in Z/2^32, the slab lists
drop only all-zero tiles, the kinfo gather implements zero padding/stride/dilation, and the
[x | xneg] image layout reproduces the -128 negate quirk (pe.cl:32-37)."""
import os

import numpy as np
import pytest

from oracle import netref, oracle as O
from tf2_amd import config as cfg, network, synth
from tests import emu_packed as emu


def _round_up(x, m):
    return (x + m - 1) // m * m


def check_net(t, q, model, images, mode, layers=None):
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(mode)
    blob = net.packed_host()
    hdr, pls = emu.parse(blob)
    R = netref.RefNet(t, q, model)
    outs = R.run(images)
    plan = R.plan
    concat_c = {}
    for L in plan:
        if L.concat >= 0:
            concat_c[L.concat] = max(concat_c.get(L.concat, 0), L.n_start + L.N)
    kinds = []
    for L in plan:
        if layers is not None and L.index not in layers:
            continue
        pl = pls[L.index]
        kinds.append(int(pl["kind"]))
        if L.ipool:
            continue
        Lx = L
        if L.src == -1 and L.C == 3 and L.k == 3 and int(pl["Cp_in"]) == 64:
            Lx, x_t = emu.first_layer_executed(L, outs[-1])       # executed as a pointwise layer over the im2col image
        elif L.src == -1:
            half = _round_up(L.C, 16)
            x_t = emu.nhwc(outs[-1], 2 * half, signed_half=half)
        elif L.src >= 0:
            S = plan[L.src]
            if S.concat >= 0:
                pytest.skip("concat sources are covered by the GPU tests")
            x_t = emu.nhwc(outs[L.src], _round_up(S.N, 16))
        else:
            cid = -(L.src + 2)
            members = [M for M in plan if M.concat == cid]
            full = np.zeros((images.shape[0], concat_c[cid]) + outs[members[0].index].shape[2:], np.int8)
            for M in members:
                full[:, M.n_start:M.n_start + M.N] = outs[M.index]
            x_t = emu.nhwc(full, _round_up(concat_c[cid], 16))
        res = outs[L.add_src] if L.add_src >= 0 else None
        if int(pl["merged_into"]) >= 0:
            continue                                        # computed (and checked) with the row in front of it
        want = outs[L.index]
        if int(pl["merge_next"]) > 0:
            # merged rows (weight_pack.cpp): this 1x1 row and the 3x3 row behind it are ONE packed 3x3 layer of both rows' channels
            import dataclasses
            Ln = plan[int(pl["merge_next"])]
            Lx = dataclasses.replace(Ln, N=L.N + Ln.N)
            want = np.concatenate([outs[L.index], outs[Ln.index]], axis=1)
        y = emu.conv_from_packed(blob, pl, Lx, x_t, res)
        # finish the layer with the oracle's post-ops and compare with the oracle's layer output
        if L.pool_en:
            y = np.stack([O.maxpool(yi, L.pool_S, L.pool_st, L.pool_pad, L.PH, L.PW) for yi in y])
        if L.endpool:
            y = np.stack([O.global_avg(yi, L.endpool_mult) for yi in y]).reshape(y.shape[0], L.N, 1, 1)
        if L.endpool and want.ndim == 4 and want.shape[2:] != (1, 1):
            want = want.reshape(want.shape[0], -1, 1, 1)
        np.testing.assert_array_equal(y, want, err_msg=f"layer {L.index} kind {int(pl['kind'])}")
    return kinds, pls


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("kind", ["float", "int8"])
def test_tiny_all_modes(mode, kind):
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 5, spread=2)
    model = synth.synth_model(t, q, 5)
    x = synth.synth_images(t, 3, 5, kind=kind)
    if kind == "int8":
        x[0, :, :2, :] = -128                     # force the negate quirk
    kinds, _ = check_net(t, q, model, x, mode)
    if mode == 0:
        assert set(kinds) <= {1, 2} and 1 in kinds
    if mode == 2:
        assert set(kinds) == {2}
        # INQ weights (7 exponents + per-channel Q): every shift layer keeps its filters as packed 4-bit codes
        assert all(int(p["fast"]) == 1 for p in _ if int(p["kind"]) == 2)


def test_squeezenet_small_image():
    t = cfg.squeezenet11_tables(image_hw=67)
    q = synth.synth_q_values(t, 6, spread=2)
    model = synth.synth_model(t, q, 6)
    x = synth.synth_images(t, 1, 6)
    check_net(t, q, model, x, 0)


def test_resnet50_selected_layers(golden_dir):
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, q, 0)
    x = synth.synth_images(t, 1, 0)
    kinds, pls = check_net(t, q, model, x, 0, layers={0, 1, 3, 4, 11, 13, 14, 46, 52, 53})
    assert set(kinds) == {1}
    # the shipped resnet50_Q has per-input-channel Q spreads up to 2 in the early layers: those
    # need a second exponent window; the uniform-Q late layers need exactly one phase
    # (layer 3 reads a two-Q tensor: one window thanks to the doubled channels, test_resnet50_doubled_channels; layer 1's input
    # carries three Q values)
    assert int(pls[1]["n_phases"]) == 2 and int(pls[3]["n_phases"]) == 1 and int(pls[46]["n_phases"]) == 1


def test_wide_shift_range_needs_more_windows():
    """Codes spanning all 32 shift amounts (far beyond INQ's 7 levels): windows of 7 cover
    them with 5 phases; wrap-around in Z/2^32 must match the oracle bit for bit."""
    t = cfg.tiny_tables(hw=8, widths=(16, 16), classes=8)
    q = synth.synth_q_values(t, 9, lo=5, hi=7, spread=0)
    rng = np.random.default_rng(9)
    plan = cfg.build_plan(t)
    model = synth.synth_model(t, q, 9)
    # overwrite layer 1's weights with powers spanning 2^0 .. 2^-14 (Get_real's whole range)
    pos = 0
    for L in plan:
        n = L.N * L.model_C * L.model_k * L.model_k
        if L.index == 1:
            e = rng.integers(0, 15, n)
            model[pos:pos + n] = np.ldexp(rng.choice([-1.0, 1.0], n), -e).astype(np.float32)
        pos += n + (L.N if L.bias_en else 0) + (4 * L.N + 1 if L.bn_en else 0)
    x = synth.synth_images(t, 2, 9)
    kinds, pls = check_net(t, q, model, x, 0)
    assert int(pls[1]["n_phases"]) >= 2
    # the shift kernel: 15 exponents do not fit the 4-bit form (7 per (row, channel)), that layer keeps int32 weights
    kinds, pls = check_net(t, q, model, x, 2)
    assert int(pls[1]["kind"]) == 2 and int(pls[1]["fast"]) == 0 and int(pls[0]["fast"]) == 1


@pytest.mark.parametrize("kind", ["float", "int8"])
def test_resnet50_first_layer_stem_image(golden_dir, kind):
    """The x-only weight tiles of conv_stem.hip: signed window values on one copy of x, plus the x = -128 correction
    derived from them, give the oracle's first layer -- with images that hold -128 (uniform int8) and without."""
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, q, 0)
    x = synth.synth_images(t, 1, 3, kind=kind)
    if kind == "int8":
        x[0, :, 5:9, :] = -128
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    blob = net.packed_host()
    _, pls = emu.parse(blob)
    assert int(pls[0]["off_w2"]) != 0
    R = netref.RefNet(t, q, model)
    outs = R.run(x)
    L = R.plan[0]
    assert (outs[-1] == -128).any() == (kind == "int8")
    y = emu.conv_stem_from_packed(blob, pls[0], L, outs[-1])
    y = np.stack([O.maxpool(yi, L.pool_S, L.pool_st, L.pool_pad, L.PH, L.PW) for yi in y])
    np.testing.assert_array_equal(y, outs[0])


def test_resnet50_wide_tile_alternatives(golden_dir):
    """Layers with >= 256 output channels on the 14x14 / 7x7 maps are packed twice: 64-row tiles (small batches, split-K) and
    128-row tiles (net.hip picks them when their grid fills the chip, or earlier when batches are in flight).  The alternative
    entry must compute the same layer: a 1x1 expand, a 3x3 and a 1x1 reduce."""
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, q, 0)
    x = synth.synth_images(t, 1, 0)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    blob = net.packed_host()
    _, pls = emu.parse(blob)
    alts = emu.parse_alt(blob)
    have = [i for i in range(len(alts)) if int(alts[i]["kind"]) == 1]
    wide = [i for i in have if int(alts[i]["TM"]) == 128]
    narrow = [i for i in have if int(alts[i]["TM"]) == 64]         # 28x28 layers -- and, since round 6, row 1 on the 56x56 maps (it then shares its batch-1 launch with row 2): 64-row tiles for the grids of batch 1-2
    assert wide and narrow and len(wide) + len(narrow) == len(have)
    assert all(int(pls[i]["TM"]) == 64 and int(pls[i]["Np"]) >= 256 for i in wide)
    assert all(int(pls[i]["TM"]) == 128 for i in narrow)
    R = netref.RefNet(t, q, model)
    outs = R.run(x)
    assert all(R.plan[i].OH * R.plan[i].OW >= 16 for i in wide) and all(R.plan[i].OH in (28, 56) for i in narrow) and 1 in narrow
    k3 = [j for j in wide if R.plan[j].k == 3]
    n3 = [j for j in narrow if R.plan[j].k == 3]
    for i in (wide[0], k3[0], k3[-1], [j for j in wide if not R.plan[j].endpool][-1], narrow[0], n3[0], n3[-1], narrow[-1]):
        L = R.plan[i]
        S = R.plan[L.src]
        x_t = emu.nhwc(outs[L.src], _round_up(S.N, 16))
        res = outs[L.add_src] if L.add_src >= 0 else None
        y = emu.conv_from_packed(blob, alts[i], L, x_t, res)
        np.testing.assert_array_equal(y, outs[i], err_msg=f"layer {i} wide-tile alternative")


def test_resnet50_doubled_channels(golden_dir):
    """Tensors inside the bottlenecks whose channels carry exactly two Q values store the higher-Q channels as 2x - 128; their
    consumers are packed with those channels' weights one exponent lower, 64 * sum(w) in the bias and -128 as the pad value --
    and need ONE exponent window where they needed two.  The packed consumer must still compute the oracle's layer."""
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, q, 0)
    x = synth.synth_images(t, 1, 0)
    kinds, pls = check_net(t, q, model, x, 0, layers={3, 4, 6, 7, 13, 14, 16, 17, 26, 27, 28, 29, 30})
    producers = [i for i in range(len(pls)) if int(pls[i]["off_dbl"])]
    consumers = [i for i in range(len(pls)) if int(pls[i]["off_pad"])]
    assert set(producers) == {2, 5, 6, 8, 12, 13, 15, 16, 18, 19, 22, 25, 26, 28, 29}, producers
    assert set(consumers) == {3, 6, 7, 9, 13, 14, 16, 17, 19, 20, 23, 26, 27, 29, 30}, consumers
    assert all(int(pls[i]["n_phases"]) == 1 for i in consumers)
    assert int(pls[28]["fast"]) == 2        # a producer off the FAST proof (SEMI form): the -128 rides in its rows' shift word




def test_fc_layers_as_4bit_codes():
    """PackLayer::fc4 (weight_pack.cpp, round 5): conv_fc layers keep their filters as 4-bit codes {sign, exponent code} in the device
    image -- per-row tables for two exponent windows and two input-channel classes, the class of every K position -- and the kernel's
    expansion (tests/emu_packed.py fc4_tiles = conv_fc.hip fc4_expand) reproduces the oracle layer by layer.  A 64 x 64 VGG16: fc6 is a
    2 x 2 window over 512 channels (32 K slabs, two-window, two classes with spread-1 Q values), fc7 reads fc6's DOUBLED channels (one
    class).  The image holds no int8 tiles for those rows: its size drops accordingly; fc4=0 keeps the window tiles (same results)."""
    os.environ["TF2_AMD_TEST"] = "1"; os.environ["TF2_AMD_OPTS"] = "fc_min=8"       # (the 64 x 64 network's fc6 has 32 K slabs; default: from 64 on)
    t = cfg.vgg16_tables(64, 40)
    q = synth.synth_q_values(t, 4, spread=1)
    model = synth.synth_model(t, q, 4)
    x = synth.synth_images(t, 2, 4)
    kinds, pls = check_net(t, q, model, x, 0, layers={13, 14})
    assert int(pls[13]["fc4"]) == 1 and int(pls[14]["fc4"]) == 1 and int(pls[15]["fc4"]) == 0      # (the last row stores the logits itself: int8)
    assert int(pls[13]["n_cls"]) == 2 and int(pls[13]["dual"]) == 1
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    size4 = net.packed_host().size
    os.environ["TF2_AMD_OPTS"] = "fc4=0,fc_min=8"
    try:
        net8 = network.NetWork(t)
        net8.Quantization(synth.q_text(q)); net8.LoadModel(model); net8.Pack(0)
        size8 = net8.packed_host().size
        _, pls8 = emu.parse(net8.packed_host())
        assert int(pls8[13]["fc4"]) == 0
    finally:
        os.environ["TF2_AMD_OPTS"] = "fc_min=8"
    # fc6: 2 windows x 4096 x 2048 int8 -> 4096 x 2048 / 2; fc7: 4096 x 4096 -> / 2
    saved = 2 * 4096 * 2048 - 4096 * 2048 // 2 + 4096 * 4096 - 4096 * 4096 // 2
    assert abs((size8 - size4) - saved) < 0.02 * saved, (size8, size4, saved)
    # three input-channel classes (spread 2 on fc6's input): the layer keeps its int8 tiles
    q3 = synth.synth_q_values(t, 4, spread=2)
    net3 = network.NetWork(t)
    net3.Quantization(synth.q_text(q3)); net3.LoadModel(synth.synth_model(t, q3, 4)); net3.Pack(0)
    _, pls3 = emu.parse(net3.packed_host())
    check_net(t, q3, synth.synth_model(t, q3, 4), x, 0, layers={13, 14})
    assert int(pls3[13]["fc4"]) == 0
    del os.environ["TF2_AMD_OPTS"]; del os.environ["TF2_AMD_TEST"]


def test_whole_window_layer_on_padded_channels_keeps_int8_tiles():
    """Round-5 advice: the pack-time condition of the 4-bit form (weight_pack.cpp PackLayer::fc4) was weaker than the run-time one
    (Net::fc_at) -- a 3 x 3 whole-window layer on a 3 x 3 x 500 tensor (Cp 512, 72 K slabs) was packed as codes that no kernel could
    run and failed to PLAN at every batch.  Both now want C == Cp_in (whole 64-channel slabs of an unpadded tensor): C = 500 keeps its
    int8 tiles and plans on the split-K kernel; the same net with C = 512 takes the codes and conv_fc.  No device needed."""
    os.environ["TF2_AMD_TEST"] = "1"; os.environ["TF2_AMD_OPTS"] = "fc_min=8"
    try:
        for C, want_fc4 in ((500, 0), (512, 1)):
            b = cfg._B(f"fcpad{C}", image=(3, 12, 12), first_filter=3)
            cur = b.conv(-1, 3, 12, 12, 64, 3, 1, 1, relu=1, pool=(2, 2, 0, 6, 6), bias=1, bn=0)
            cur = b.conv(cur, 64, 6, 6, C, 3, 1, 1, relu=1, pool=(2, 2, 0, 3, 3), bias=1, bn=0)
            cur = b.conv(cur, C, 3, 3, 128, 3, 1, 0, relu=1, bias=1, bn=0)          # the whole-window layer: 9 * Cp / 64 = 72 slabs
            b.conv(cur, 128, 1, 1, 16, 1, 1, 0, relu=0, bn=0, bias=1)
            t = b.tables()
            q = synth.synth_q_values(t, 3, spread=1)
            model = synth.synth_model(t, q, 3)
            x = synth.synth_images(t, 2, 3)
            _, pls = check_net(t, q, model, x, 0, layers={2})
            assert int(pls[2]["fc4"]) == want_fc4, (C, pls[2]["fc4"])
            net = network.NetWork(t)
            net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
            for batch in (1, 32, 64):
                for conc in (0, 1):
                    rows = net.describe_launches(batch, conc)
                    kern = [r["kernel"] for r in rows if r["layer"] == 2]
                    assert kern and (kern[0].startswith("fc4") == bool(want_fc4)), (C, batch, kern)
    finally:
        del os.environ["TF2_AMD_OPTS"]; del os.environ["TF2_AMD_TEST"]
