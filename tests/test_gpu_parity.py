"""GPU parity proper: the HIP path, called through the C ABI, against the oracle on the same
seeded inputs -- bit-exact, layer by layer (the reference's own per-layer Verify practice,
network_helper.cpp:36-75) and on the final logits / top-5 (Evaluation)."""
import os

import numpy as np
import pytest

from oracle import netref
from tf2_amd import config as cfg, network, synth

from tests.conftest import set_opts  # noqa: E402

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch


class Rig:
    def __init__(self, tables, q, model, mode):
        torch = _torch()
        self.t, self.q, self.model = tables, q, model
        self.net = network.NetWork(tables)
        self.net.Init(model, synth.q_text(q), device="cuda:0", pack_mode=mode)
        self.runner = network.Runner(None, self.net)
        self.ref = netref.RefNet(tables, q, model)

    def run(self, images, keep_all=True):
        torch = _torch()
        x = torch.from_numpy(np.ascontiguousarray(images)).to("cuda:0")
        logits = self.runner.run_batch(x, keep_all=keep_all)
        torch.cuda.synchronize()
        return logits.cpu().numpy()

    def check_all_layers(self, images, layers=None):
        got_logits = self.run(images, keep_all=True)
        outs = self.ref.run(images)
        B = images.shape[0]
        np.testing.assert_array_equal(self.runner.read_layer(-1, B), outs[-1], err_msg="network input (prep kernel)")
        for L in self.ref.plan:
            if layers is not None and L.index not in layers:
                continue
            got = self.runner.read_layer(L.index, B)
            np.testing.assert_array_equal(got, outs[L.index], err_msg=f"layer {L.index}")
        want = self.ref.logits(outs)
        np.testing.assert_array_equal(got_logits, want)
        for b in range(B):
            lab_g, _ = network.Evaluation(b, self.net.q[len(self.ref.plan)], got_logits)
            lab_o, _ = self.ref.top5(want[b])
            assert lab_g == lab_o.tolist()
        return got_logits


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("kind", ["float", "int8"])
def test_tiny_every_layer(mode, kind):
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 5, spread=2)
    model = synth.synth_model(t, q, 5)
    x = synth.synth_images(t, 5, 5, kind=kind)
    if kind == "int8":
        x[0, :, :2, :] = -128          # the negate quirk of pe.cl:32-37
    Rig(t, q, model, mode).check_all_layers(x)


@pytest.mark.parametrize("mode", [0, 2])
def test_squeezenet_concat(mode):
    t = cfg.squeezenet11_tables(image_hw=99)
    q = synth.synth_q_values(t, 6, spread=2)
    model = synth.synth_model(t, q, 6)
    x = synth.synth_images(t, 3, 6)
    Rig(t, q, model, mode).check_all_layers(x)


@pytest.mark.parametrize("nofast", ["0", "1"])     # (the shift kernels' "fast" form is the 4-bit packed one: no4bit = 0 / 1)
def test_shift_kernel_wave_split_fc(nofast, monkeypatch):
    """conv_shift_fc_kernel (a 1x1 layer on a handful of pixels with a long channel walk: sixteen waves take every sixteenth
    16-channel chunk, partial sums added through LDS) against the oracle -- unsigned input with 4-bit-packed and int32 filters
    (mode 2: every layer on the shift kernels), and the signed input of SqueezeNet's classifier (default mode)."""
    set_opts(monkeypatch, no4bit=nofast)
    t = cfg.tiny_tables(hw=12, widths=(32, 512), classes=100)
    q = synth.synth_q_values(t, 9, spread=2)
    model = synth.synth_model(t, q, 9)
    rig = Rig(t, q, model, 2)
    names = [r["kernel"] for r in rig.net.describe_launches(3, 0)]
    assert names[-1] == "conv_shift_fc_kernel", names
    rig.check_all_layers(synth.synth_images(t, 3, 9))
    t = cfg.squeezenet11_tables(image_hw=67)
    q = synth.synth_q_values(t, 6, spread=2)
    rig = Rig(t, q, synth.synth_model(t, q, 6), 0)
    assert rig.net.describe_launches(5, 0)[-1]["kernel"] == "conv_shift_fc_kernel"
    rig.check_all_layers(synth.synth_images(t, 5, 6))


def test_first_3x3_layer_plain_form(monkeypatch):
    """im2col0=0: a 3x3 first layer on the 3-channel image in its plain form (nine taps of [x | xneg] gathered by the ring
    kernel) instead of the default pointwise layer over the im2col image -- both bit-exact, -128 pixels included, and the input
    tensor read back as the quantised image either way."""
    for env in ("0", "1"):
        set_opts(monkeypatch, im2col0=env)
        t = cfg.tiny_tables()
        q = synth.synth_q_values(t, 5, spread=2)
        x = synth.synth_images(t, 3, 5, kind="int8")
        x[0, :, :2, :] = -128
        rig = Rig(t, q, synth.synth_model(t, q, 5), 0)
        first = rig.net.describe_launches(3, 0)[0]["kernel"]
        assert ("im2col" in first) == (env == "1"), first
        rig.check_all_layers(x)
        t = cfg.squeezenet11_tables(image_hw=67)               # stride 2, no padding
        q = synth.synth_q_values(t, 6, spread=2)
        Rig(t, q, synth.synth_model(t, q, 6), 0).check_all_layers(synth.synth_images(t, 2, 6), layers={0, 1, 2})


@pytest.mark.parametrize("c3", ["1", "0"])
def test_3x3_layers_from_an_lds_resident_halo_tile(c3, monkeypatch):
    """conv_c3.hip (a 3x3 / 1 / pad 1 layer with the input's halo tile streamed through LDS once: VGG16 / SSD300's body) against the
    oracle on every layer of a 64 x 64 VGG16 (maps 64 / 32 / 16: 1-8 channel slabs, 64- and 128-channel blocks, one- and two-window
    rows, partial tiles at the map's edge; the 8 x 8 and 4 x 4 maps stay on the ring / split-K kernels), and the same network with the
    kernel switched off."""
    set_opts(monkeypatch, c3=c3)
    set_opts(monkeypatch, c3_min="1")
    set_opts(monkeypatch, c3_min256="1")           # 256-channel blocks (two row tiles per wave) wherever a layer allows them
    set_opts(monkeypatch, c3_w9="2")               # the weights-resident kernel for the 64 -> 64 layer although a block walks few tiles
    t = cfg.vgg16_tables(64, 10)
    q = synth.synth_q_values(t, 7, spread=2)
    rig = Rig(t, q, synth.synth_model(t, q, 7), 0)
    names = [r["kernel"] for r in rig.net.describe_launches(3, 0)]
    assert any("conv_c3" in n for n in names) == (c3 == "1"), names
    x = synth.synth_images(t, 3, 7)
    rig.check_all_layers(x)
    # one Q value per tensor: every layer one-window -- the 64 -> 64 layer then takes the weights-resident kernel (conv_c3_w9: nine
    # fragments in registers, the input three tiles ahead), the 256- and 512-channel ones 256-channel blocks
    q1 = synth.synth_q_values(t, 0, spread=1)
    rig1 = Rig(t, q1, synth.synth_model(t, q1, 0), 0)
    names1 = [r["kernel"] for r in rig1.net.describe_launches(5, 0)]
    assert any("conv_c3_w9" in n for n in names1) == (c3 == "1") and any("<256 channels" in n for n in names1) == (c3 == "1"), names1
    rig1.check_all_layers(synth.synth_images(t, 5, 2))
    t = cfg.tiny_tables(hw=40, widths=(64, 128), classes=10)      # a map of 20 x 20 (one tile with a ragged last column tile), residual net
    q = synth.synth_q_values(t, 3, spread=2)
    Rig(t, q, synth.synth_model(t, q, 3), 0).check_all_layers(synth.synth_images(t, 2, 3))


@pytest.mark.parametrize("form", ["codes", "tiles", "off"])
def test_whole_window_layers_as_a_weight_stream(form, monkeypatch):
    """conv_fc.hip (a layer whose input is one filter window per image -- VGG16's fc6 / fc7 -- as a weight stream over the whole chip:
    output channels x K slices, int32 partial sums through the workspace's scratch area, a finishing pass) against the oracle: a 64 x 64
    VGG16 (fc6 = 2 x 2 x 512 -> 4096 on a 2 x 2 map, fc7 1 x 1: 32 and 64 slabs, one and several K slices, one- and two-window rows).
    codes: the default -- the filters stay 4-BIT CODES in HBM and are expanded in registers (PackLayer::fc4, fc4_partial_kernel; two
    input-channel classes with spread-1 Q values, one behind doubled channels; spread 2 = three classes keeps int8 tiles), any batch in
    chunks of 32 images; tiles: fc4=0, int8 window tiles (batch <= 32, else the split-K kernel); off: the split-K kernel."""
    set_opts(monkeypatch, fc="0" if form == "off" else "1", fc4="1" if form == "codes" else "0", fc_min="8")
    t = cfg.vgg16_tables(64, 10)
    for seed, spread in ((7, 2), (0, 1)):
        q = synth.synth_q_values(t, seed, spread=spread)
        rig = Rig(t, q, synth.synth_model(t, q, seed), 0)
        names = [r["kernel"] for r in rig.net.describe_launches(5, 0)]
        assert any("fc_partial" in n or "fc4_partial" in n for n in names) == (form != "off"), names
        assert any("fc4_partial" in n for n in names) == (form == "codes" and spread == 1), names
        batches = (1, 5, 32, 33, 70) if form == "codes" else (1, 5, 32)
        if form == "codes" and spread == 1:
            assert sum("fc4_partial" in r["kernel"] for r in rig.net.describe_launches(70, 1)) == 2       # three chunks of 32 images, still conv_fc
        for b in batches:
            rig.check_all_layers(synth.synth_images(t, b, seed + b), layers={12, 13, 14, 15})


def test_vgg_small_bias_and_2x2_pools():
    t = cfg.vgg16_tables(32, 10)
    q = synth.synth_q_values(t, 7)
    model = synth.synth_model(t, q, 7)
    x = synth.synth_images(t, 2, 7)
    Rig(t, q, model, 0).check_all_layers(x)


def test_wide_shift_range_phases():
    t = cfg.tiny_tables(hw=8, widths=(16, 16), classes=8)
    q = synth.synth_q_values(t, 9, lo=5, hi=7, spread=0)
    rng = np.random.default_rng(9)
    model = synth.synth_model(t, q, 9)
    pos = 0
    for L in cfg.build_plan(t):
        n = L.N * L.model_C * L.model_k * L.model_k
        if L.index == 1:
            model[pos:pos + n] = np.ldexp(rng.choice([-1.0, 1.0], n), -rng.integers(0, 15, n)).astype(np.float32)
        pos += n + (L.N if L.bias_en else 0) + (4 * L.N + 1 if L.bn_en else 0)
    x = synth.synth_images(t, 2, 9)
    Rig(t, q, model, 0).check_all_layers(x)


@pytest.fixture(scope="module")
def r50(golden_dir):
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, q, 0)
    return t, q, model


@pytest.fixture(scope="module")
def r50_rig(r50):
    return Rig(*r50, 0)


def test_resnet50_every_layer_batch2(r50_rig):
    x = synth.synth_images(r50_rig.t, 2, 0)
    r50_rig.check_all_layers(x)


def test_resnet50_shipped_test_image(r50_rig, golden_dir):
    """The reference's own fixture image (test_images/resnet50_data_label_100.bin) through the
    seeded synthetic weights: logits and top-5 identical to the oracle."""
    img = np.fromfile(os.path.join(golden_dir, "resnet50_data_label_100.bin"), dtype=np.float32).reshape(1, 3, 224, 224)
    r50_rig.check_all_layers(img, layers={0, 10, 52, 53})


def test_resnet50_int8_input_with_minus128(r50_rig):
    x = synth.synth_images(r50_rig.t, 1, 3, kind="int8")
    x[0, :, 100:120, 100:120] = -128
    r50_rig.check_all_layers(x, layers={0, 1, 4, 53})


def test_resnet50_stem_blocks_with_and_without_minus128(r50_rig):
    """conv_stem.hip decides per block (7 output rows of one image) whether the x = -128 correction runs: a batch in
    which one image has a small saturated patch, one is saturated everywhere and two have no -128 at all."""
    x = synth.synth_images(r50_rig.t, 4, 21)
    x[1, :, 60:64, 150:170] = -1000.0            # clamps to -128 in a few row bands of image 1
    x[2] = -1000.0
    outs = r50_rig.ref.run(x)
    q_in = outs[-1]
    assert (q_in[1] == -128).any() and not (q_in[0] == -128).any() and not (q_in[3] == -128).any()
    r50_rig.check_all_layers(x, layers={0, 53})


@pytest.mark.parametrize("conc", ["1", "0"])
def test_minus128_flags_from_the_input_preparation(r50, monkeypatch, conc):
    """Round 6: where a step starts as prep_rewrite3_rows_kernel | conv_stem_pool_kernel | conv_bfirst_kernel (batch >= 12, both launch plans), the input
    preparation reports per image whether a quantised element is -128 (runner.cpp:158-164 clamps to it; pe.cl:32-37 negates (int8)(-x) there), the
    stem's blocks read their image's word instead of scanning their input tile, and conv_bfirst clears the words for the next step.  A sequence of
    steps on ONE workspace -- no -128 anywhere, a small saturated patch in one image, another image saturated everywhere, clean again, float and
    int8 sources -- every step's conv1 output and logits against the oracle, and against the library with the hand-over off (q128=0)."""
    set_opts(monkeypatch, alt_conc=conc)
    rig = Rig(*r50, 0)
    t = rig.t
    B = 16
    clean = synth.synth_images(t, B, 211)
    patch = clean.copy(); patch[5, :, 60:64, 150:170] = -1000.0
    full = clean.copy(); full[9] = -1000.0; full[0, 1, 0, 0] = -1000.0
    q8 = synth.synth_images(t, B, 212, kind="int8"); q8[3, 2, 223, 220:] = -128
    got = []
    for x in (clean, patch, clean, full, q8, clean):
        logits = rig.run(x, keep_all=True).copy()
        outs = rig.ref.run(x[[0, 3, 5, 9]])
        np.testing.assert_array_equal(rig.runner.read_layer(0, B)[[0, 3, 5, 9]], outs[0])
        np.testing.assert_array_equal(logits[[0, 3, 5, 9]], rig.ref.logits(outs))
        got.append(logits)
    np.testing.assert_array_equal(got[0], got[2]); np.testing.assert_array_equal(got[0], got[5])
    set_opts(monkeypatch, q128="0")
    plain = Rig(*r50, 0)
    for x, want in zip((clean, patch, full, q8), (got[0], got[1], got[3], got[4])):
        np.testing.assert_array_equal(plain.run(x, keep_all=True), want)


def test_resnet50_full_batch32_logits_and_properties(r50_rig):
    """BASELINE batch: all 32 logits rows against the oracle; batch invariance (row i of the
    batch run == the same image run alone); determinism."""
    x = synth.synth_images(r50_rig.t, 32, 11)
    got = r50_rig.run(x, keep_all=False)
    outs = r50_rig.ref.run(x)
    np.testing.assert_array_equal(got, r50_rig.ref.logits(outs))
    again = r50_rig.run(x, keep_all=False)
    np.testing.assert_array_equal(got, again)
    alone = r50_rig.run(x[7:8], keep_all=False)
    np.testing.assert_array_equal(alone[0], got[7])


def test_resnet50_north_star_split_mode1(r50):
    """3x3 convs on the shift-accumulate VALU kernel, 1x1 on int8 MFMA."""
    rig = Rig(*r50, 1)
    x = synth.synth_images(rig.t, 1, 2)
    rig.check_all_layers(x, layers={0, 3, 4, 13, 26, 45, 52, 53})


@pytest.mark.parametrize("sk", ["1", "2"])
def test_split_k_kernel_forced_and_disabled(sk, monkeypatch):
    """conv_mfma_sk.hip (four waves split the slab list) forced for every 64-row layer, and disabled:
    both bit-exact, including layers with fewer slabs than waves and several Horner phases."""
    set_opts(monkeypatch, sk=sk)
    t = cfg.squeezenet11_tables(image_hw=67)
    q = synth.synth_q_values(t, 12, spread=2)
    model = synth.synth_model(t, q, 12)
    Rig(t, q, model, 0).check_all_layers(synth.synth_images(t, 2, 12))
    t = cfg.tiny_tables(hw=20, widths=(64, 128), classes=100)
    q = synth.synth_q_values(t, 8, spread=2)
    model = synth.synth_model(t, q, 8)
    Rig(t, q, model, 0).check_all_layers(synth.synth_images(t, 3, 8))


def test_generic_requant_forced(r50, monkeypatch):
    """nofast=1 at pack time: every layer takes the 6-instruction wrap-exact requantisation instead of the
    range-proven 3-instruction one (most synthetic ResNet-50 layers qualify for the latter); same bits."""
    set_opts(monkeypatch, nofast="1")
    rig = Rig(*r50, 0)
    rig.check_all_layers(synth.synth_images(rig.t, 2, 23), layers={1, 2, 3, 4, 11, 13, 26, 28, 45, 52, 53})
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 5, spread=2)
    model = synth.synth_model(t, q, 5)
    Rig(t, q, model, 0).check_all_layers(synth.synth_images(t, 2, 5))


def test_single_window_packing_forced(r50, monkeypatch):
    """nodual=1 at pack time: two-phase layers keep the Horner form (one exponent window per entry, accumulators
    shifted at the phase boundary) instead of the default dual-window entries; conv_mfma2 and the split-K kernel."""
    set_opts(monkeypatch, nodual="1")
    rig = Rig(*r50, 0)
    rig.check_all_layers(synth.synth_images(rig.t, 2, 29), layers={0, 1, 2, 3, 4, 11, 13, 24, 26, 27, 29, 47, 53})


def test_pointwise_register_kernel_on_and_off(r50, monkeypatch):
    """conv_pw.hip (weights in registers, activations global -> register -> MFMA, pixel tiles streamed per wave) is the
    default for dense 1x1 stride-1 layers with K <= 128; pw=0 sends them back to conv_mfma2.  Ragged pixel
    counts (batch 3 and 5), with / without residual, one and two K slabs, single- and dual-window packing."""
    set_opts(monkeypatch, pw_slabs="2")      # also the two-slab instantiations (default: one slab only)
    set_opts(monkeypatch, pw_minpix="0")     # ... and at these small pixel counts (default: from 8192 pixels on)
    rig = Rig(*r50, 0)
    pw_layers = {1, 2, 4, 7, 10, 14, 17, 20, 23}
    for b, seed in ((3, 51), (5, 52)):
        rig.check_all_layers(synth.synth_images(rig.t, b, seed), layers=pw_layers | {3, 12, 53})
    set_opts(monkeypatch, pw="0")
    rig.net.reload_options()
    rig.check_all_layers(synth.synth_images(rig.t, 3, 53), layers=pw_layers)
    set_opts(monkeypatch, pw=None)
    set_opts(monkeypatch, nodual="1")       # single-window packing: two-phase layers are not eligible, one-phase are
    t = cfg.tiny_tables(hw=20, widths=(64, 128), classes=100)
    q = synth.synth_q_values(t, 8, spread=0)
    model = synth.synth_model(t, q, 8)
    Rig(t, q, model, 0).check_all_layers(synth.synth_images(t, 3, 8))


def test_first_layer_generic_path_and_wide_tile_alternatives(r50, monkeypatch):
    """Two run-time choices against the oracle, every layer: (1) stem=0 -- the first layer on the generic ring kernel
    over [x | xneg] instead of conv_stem.hip on x alone; (2) alt_min=0 -- the 128-row alternatives of the layers with
    >= 1024 output channels (normally taken only when their grid fills the chip, i.e. from batch 32 on) at a ragged batch of 3."""
    set_opts(monkeypatch, stem="0")
    set_opts(monkeypatch, alt_min="0")
    rig = Rig(*r50, 0)
    x = synth.synth_images(rig.t, 3, 71)
    x[1, :, 10:14, :] = -1000.0                      # clamps to -128: the negate quirk through the xneg half
    rig.check_all_layers(x)
    set_opts(monkeypatch, stem=None)
    set_opts(monkeypatch, alt_min="1000000")  # never: the 64-row tiles / split-K kernel for every small-map layer
    rig.net.reload_options()
    rig.check_all_layers(x, layers={0, 24, 27, 43, 46, 52, 53})


def test_doubled_channels_on_and_off(r50, monkeypatch):
    """Internal two-Q tensors store their higher-Q channels as 2x - 128 (one exponent window for the consumers,
    weight_pack.cpp): the default run reads them back through read_layer's inverse in every other test; here the plain form
    (nodbl=1 at pack time) on the same images, every layer, and both against the oracle.  The images saturate some
    3x3 borders' neighbourhoods so that padded taps (pad value -128 on doubled channels) matter."""
    x = synth.synth_images(r50[0], 3, 81)
    x[1, :, :8, :] = 150.0
    x[2, :, :, -6:] = -120.0
    Rig(*r50, 0).check_all_layers(x)
    set_opts(monkeypatch, nodbl="1")
    rig = Rig(*r50, 0)
    rig.check_all_layers(x, layers={2, 3, 4, 6, 7, 12, 13, 14, 16, 17, 26, 27, 53})


def test_fused_bottleneck_pairs(r50, monkeypatch):
    """conv_bneck.hip: branch2b (3x3 / stride 1) + branch2c (1x1 expand, residual, ReLU) of the stride-1 bottlenecks in one
    launch (C = 64 / 128 / 256; halo tile and intermediate tile in LDS, weights from registers): every layer with keep_all
    (the intermediate map is then also stored) at ragged batches, the logits of plain runs, and the unfused path
    (nofuse=1 at pack time) on the same inputs.  bneck_min=1: the fused launch also for the small grids of
    these batches (by default a pair runs fused only when it has at least 256 row bands, i.e. blocks)."""
    set_opts(monkeypatch, bneck_min="1")
    rig = Rig(*r50, 0)
    rig.check_all_layers(synth.synth_images(rig.t, 3, 61))
    x = synth.synth_images(rig.t, 5, 62)
    want = rig.ref.logits(rig.ref.run(x))
    np.testing.assert_array_equal(rig.run(x, keep_all=False), want)
    set_opts(monkeypatch, nofuse="1")
    rig2 = Rig(*r50, 0)
    np.testing.assert_array_equal(rig2.run(x, keep_all=False), want)
    rig2.check_all_layers(synth.synth_images(rig.t, 2, 63), layers={3, 4, 16, 17, 32, 33, 52, 53})


def test_googlenet_ipool_concat_every_layer(golden_dir):
    """The reference's second shipped network (googlenet.h tables + shipped googlenet_Q): 67 table rows with independent
    pooling rows (kIpoolEnable), four-way concat slices (kNStart/kBranchTail/kConcatLayer, extra Q rows), 5x5 convs, the
    7x7 conv1 rewrite and LRN-free inception blocks -- every layer against the oracle."""
    import json
    g = json.load(open(os.path.join(golden_dir, "tables_googlenet.json")))
    t = cfg.NetTables(g)
    t.setdefault("xConv1Rewrite", 1)
    qv = np.loadtxt(os.path.join(golden_dir, "googlenet_Q"), dtype=np.int32)
    model = synth.synth_model(t, qv, 7)
    rig = Rig(t, qv, model, 0)
    assert sum(L.ipool for L in rig.ref.plan) == 9
    rig.check_all_layers(synth.synth_images(t, 2, 3))


def test_squeezenet_227_batch32_properties():
    """BASELINE configs[1] at full size (SqueezeNet 1.1, 32 x 3 x 227 x 227): the oracle checks the first two images in
    full; the rest through size-independent properties -- every image's logits equal those of the same image run
    alone-in-a-pair, and a batch permutation permutes the logits."""
    t = cfg.squeezenet11_tables()
    q = synth.synth_q_values(t, 21, spread=2)
    model = synth.synth_model(t, q, 21)
    rig = Rig(t, q, model, 0)
    x = synth.synth_images(t, 32, 77)
    got = rig.run(x, keep_all=False)
    want2 = rig.ref.logits(rig.ref.run(x[:2]))
    np.testing.assert_array_equal(got[:2], want2)
    perm = np.random.default_rng(5).permutation(32)
    np.testing.assert_array_equal(rig.run(x[perm], keep_all=False), got[perm])
    np.testing.assert_array_equal(rig.run(x[30:32], keep_all=False), got[30:32])


def test_vgg16_full_size_batch2_every_layer():
    """BASELINE configs[3]'s network at full size (VGG16, 224x224: 13 3x3 convs on large maps with 2x2 pools, fc6 as a
    7x7 convolution over 512 channels = 25088-deep K, fc7/fc8): every layer against the oracle at batch 2."""
    t = cfg.vgg16_tables()
    q = synth.synth_q_values(t, 1, spread=1)
    model = synth.synth_model(t, q, 1)
    rig = Rig(t, q, model, 0)
    rig.check_all_layers(synth.synth_images(t, 2, 6))


def test_input_quantisation_ties_and_extremes(r50_rig):
    """prep_input_kernel restates runner.cpp:158-164 without double precision: exact .5 ties of both signs, values
    straddling the int8 clamp, denormals, -0.0, and |x| >= 2^31 / inf / NaN (x86 cvttsd2si -> INT_MIN -> -128 in the
    reference binary and the oracle) must come out identically."""
    rig = r50_rig
    rng = np.random.default_rng(99)
    x = synth.synth_images(rig.t, 2, 61)
    q0 = int(rig.net.q[0][0])                      # runtime (negated) Q of image channel 0
    scale = float(2.0 ** q0) if q0 > 0 else float(2.0 ** q0)
    flat = x.reshape(-1)
    ties = (rng.integers(-140, 141, size=20000).astype(np.float32) + 0.5) * np.float32(scale)
    flat[:20000] = ties
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, 127.49, 127.5, -128.5, -128.51, 2147483648.0, -2147483648.0,
                         4e9, -4e9, 1e30, -1e30, np.inf, -np.inf, np.nan, 2147483520.0, -2147483520.0], np.float32)
    flat[20000:20000 + specials.size] = specials
    flat[30000:40000] = (rng.uniform(-200, 200, 10000)).astype(np.float32)
    rig.run(x, keep_all=True)
    want = rig.ref.run(x)[-1]
    np.testing.assert_array_equal(rig.runner.read_layer(-1, 2), want)


@pytest.mark.parametrize("blocks,kmax", [("8", "8"), ("64", "8"), ("64", "2")])
def test_split_k_over_blocks_of_the_small_grids(r50, monkeypatch, blocks, kmax):
    """conv_mfma_sk.hip KSP (round 6): split-K launches of a few blocks (batch 1: the 7 x 7 maps' 3x3 and 2048 -> 512 rows, eight blocks that each stream
    150-300 KB of weights) split K over BLOCKS as well -- ks_parts blocks per output tile leave their 64 x 64 int32 partial tiles in the scratch area
    (write-through stores, acknowledged), draw a device-scope ticket, and the block that draws the tile's last one adds the parts up (loads past its
    caches) and requantises; the ticket words are cleared by the step's first kernel.  Every layer against the oracle at batch 1 / 2 / 3 (sk_kb_blocks=64:
    the 14 x 14 and 28 x 28 rows as well, padded and two-window instantiations, 2 / 4 / 8 parts), 200 graph replays of a batch-1 step, four runners at
    once on their own workspaces, and against sk_kb=0."""
    set_opts(monkeypatch, sk_kb_blocks=blocks, sk_kb_max=kmax, sk_kb_min="2")
    rig = Rig(*r50, 0)
    mine = [r for r in rig.net.describe_launches(1, 0) if "K over" in r["kernel"]]
    assert {45, 48, 51} <= {r["layer"] for r in mine} and (blocks == "8" or len(mine) >= 12)
    for b, seed in ((1, 301), (2, 302), (3, 303)):
        rig.check_all_layers(synth.synth_images(rig.t, b, seed, kind="int8" if b == 2 else "float"))
    torch = _torch()
    x1 = synth.synth_images(rig.t, 1, 304)
    want = rig.ref.logits(rig.ref.run(x1))
    xg = torch.from_numpy(x1).to("cuda:0")
    fn = rig.runner.capture(xg, concurrency=0)
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rig.runner._logits.cpu().numpy(), want)
    runners = [network.Runner(None, rig.net) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    for i in range(100):
        for r, s in zip(runners, streams):
            with torch.cuda.stream(s):
                r.run_batch(xg, concurrency=0)
    torch.cuda.synchronize()
    for r in runners:
        np.testing.assert_array_equal(r._logits.cpu().numpy(), want)
    set_opts(monkeypatch, sk_kb="0")
    plain = Rig(*r50, 0)
    assert not any("K over" in r["kernel"] for r in plain.net.describe_launches(1, 0))
    np.testing.assert_array_equal(plain.run(x1, keep_all=False), want)


@pytest.mark.parametrize("sk8", ["0", "100000"])
def test_resnet50_split_k_forced(r50, monkeypatch, sk8):
    """4-way and 8-way in-block split-K (sk8 = largest grid that takes the 8-wave form)."""
    set_opts(monkeypatch, sk8=sk8)
    set_opts(monkeypatch, sk="1")
    rig = Rig(*r50, 0)
    rig.check_all_layers(synth.synth_images(rig.t, 2, 21), layers={26, 28, 32, 45, 47, 52, 53})


def test_runner_run_mirrors_reference_flow(r50_rig, golden_dir):
    """Runner::Run: image file -> num_images frames -> output + throughput (runner.cpp:54-198)."""
    net = r50_rig.net
    net.image_file = os.path.join(golden_dir, "resnet50_data_label_100.bin")
    net.num_images = 2
    r = network.Runner(None, net)
    r.Init()
    out = r.Run()
    assert out.shape == (2, 1000) and (out[0] == out[1]).all() and r.throughput_fps > 0
    err = network.Verify(0, os.path.join(golden_dir, "resnet50_fc1000_label_100.bin"), net.q, out, num_layer=net.num_layer)
    assert np.isfinite(err)        # synthetic weights: the number is meaningless, the plumbing is what is checked


def _stem_nopool_tables():
    b = cfg._B("stem_nopool", image=(3, 224, 224), first_filter=7, rewrite=1)
    a = b.conv(-1, 27, 114, 114, 64, 3, 1, 0, relu=1)                       # conv1 in its executed form, NO pool
    x = b.conv(a, 64, 112, 112, 64, 3, 2, 1, relu=1)                          # reads the (doubled) stem output
    y = b.conv(x, 64, 56, 56, 32, 1, 1, 0, relu=1, endpool=1, endpool_hw=56 * 56)
    b.conv(y, 32, 1, 1, 10, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


def _expand_nores_tables():
    b = cfg._B("expand_nores", image=(3, 32, 32), first_filter=3)
    a = b.conv(-1, 3, 32, 32, 64, 3, 1, 1, relu=1)
    x = b.conv(a, 64, 32, 32, 64, 3, 1, 1, relu=1)                            # 3x3 C -> C ...
    e = b.conv(x, 64, 32, 32, 256, 1, 1, 0, relu=1)                           # ... and its 1x1 expand WITHOUT a shortcut
    r = b.conv(e, 256, 32, 32, 64, 1, 1, 0, relu=1)                           # reads the (doubled) expand output
    y = b.conv(r, 64, 32, 32, 32, 3, 2, 1, relu=1, endpool=1, endpool_hw=16 * 16)
    b.conv(y, 32, 1, 1, 10, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


@pytest.mark.parametrize("which", ["stem_without_pool", "fused_expand_without_residual"])
def test_doubled_channels_written_by_conv_stem_and_by_a_fused_expand(which, monkeypatch):
    """The two producers of a "doubled" tensor (weight_pack.cpp: channels stored as 2y - 128) that ResNet-50 never exercises:
    conv_stem (a first layer without a pool) and conv_bneck's expand epilogue (an expand without a shortcut).  Their header rows
    carry the -128 and their consumers' weights are halved, so the kernels must apply the doubling -- every layer vs the oracle."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import emu_packed as emu
    set_opts(monkeypatch, bneck_min="1")
    t = _stem_nopool_tables() if which == "stem_without_pool" else _expand_nores_tables()
    q = synth.synth_q_values(t, 1, spread=1)
    rig = Rig(t, q, synth.synth_model(t, q, 1), 0)
    _, pls = emu.parse(rig.net.packed_host())
    launches = rig.net.describe_launches(4, 0)
    if which == "stem_without_pool":
        assert int(pls[0]["off_dbl"]) != 0 and any("conv_stem" in l["kernel"] for l in launches)
    else:
        assert int(pls[2]["off_dbl"]) != 0 and int(pls[2]["fused_into"]) == 1 and any("conv_bneck" in l["kernel"] for l in launches)
    rig.check_all_layers(synth.synth_images(t, 4, 3))


@pytest.mark.parametrize("chain", ["5", "2", "1"])
def test_group_launches_of_the_identity_bottlenecks(r50, monkeypatch, chain):
    """bgroup=1 (the default): the five identity bottlenecks of stage 4 (rows 28-42) and the two of stage 5 (rows 47-52, the
    second one ending in the global average) as ONE launch each, eight blocks per image meeting at epoch-tagged flags between
    the layers (conv_bgroup.hip); stage 4's five in ONE launch together (bgroup_chain=5, the default), as 2 + 2 + 1, or
    one by one.  Every layer against the oracle at batch 2 and 5, then batch-32 logits of repeated runs on the
    liveness-planned workspace."""
    set_opts(monkeypatch, bgroup="1")
    set_opts(monkeypatch, bgroup_chain=chain)
    set_opts(monkeypatch, bgroup_min7="1")          # (by default batches below 12 keep the separate launches: measured equal or faster there)
    set_opts(monkeypatch, bgroup_min14="1")
    set_opts(monkeypatch, bgroup_min28="1")
    set_opts(monkeypatch, bgroup_min56f="1")
    set_opts(monkeypatch, bfirst="1")               # (rows 1-4 on their GROUP launch: since round 6 the default takes conv_bfirst.hip in both plans)
    set_opts(monkeypatch, alt_conc="0")
    rig = Rig(*r50, 0)
    rows = rig.net.describe_launches(32, 0)
    stage45 = {"5": [28, 47], "2": [28, 34, 40, 47], "1": [28, 31, 34, 37, 40, 47, 50]}[chain]
    stage3 = [15, 18, 21] if chain == "1" else [15, 21]      # (rows 15-20 share a launch; row 21's 3x3 is a two-window layer: another instantiation)
    assert [r["layer"] for r in rows if "conv_bgroup" in r["kernel"]] == [1] + stage3 + stage45
    assert "dual reduce" in [r for r in rows if r["layer"] == 47][0]["kernel"] and "global average" in [r for r in rows if r["layer"] == (50 if chain == "1" else 47)][0]["kernel"]
    rig.check_all_layers(synth.synth_images(rig.t, 2, 71))
    rig.check_all_layers(synth.synth_images(rig.t, 5, 72))
    x = synth.synth_images(rig.t, 32, 73)
    first = rig.run(x, keep_all=False).copy()
    np.testing.assert_array_equal(first[:3], rig.ref.logits(rig.ref.run(x[:3])))
    for _ in range(10):
        np.testing.assert_array_equal(rig.run(x, keep_all=False), first)
    set_opts(monkeypatch, bgroup="0")
    plain = Rig(*r50, 0)
    np.testing.assert_array_equal(plain.run(x, keep_all=False), first)


@pytest.mark.parametrize("withhold_row", [28, 47, 15, 1])
def test_group_launch_that_cannot_meet_reports_a_status(r50, monkeypatch, withhold_row):
    """include/tf2_amd.h promises status codes, never a dead context (round-5 review: the group kernels ended a stuck meeting with
    __builtin_trap()).  The test-only option bgroup_withhold makes ONE block of every group launch leave at kernel entry without posting
    its flags: the other seven members of its image give up after bgroup_polls polls, write a report into the workspace's error word and
    return; every other image completes; the step's later launches run; tf2_net_poll_error returns TF2_ERR_GROUP once (the report is
    sticky until polled, then cleared), and the SAME handle, workspace and HIP context then run a clean step with the oracle's logits.
    One case per group kernel (14 x 14 chain, 7 x 7 chain, 28 x 28, the first 56 x 56 bottleneck: the block index picks a member of image 0
    or 1 of each)."""
    from tf2_amd._lib import Tf2Error
    torch = _torch()
    set_opts(monkeypatch, bgroup="1", alt_conc="0", bgroup_min7=1, bgroup_min14=1, bgroup_min28=1, bgroup_min56f=1, bfirst="1")
    rig = Rig(*r50, 0)
    x = synth.synth_images(rig.t, 9, 75)
    want = rig.ref.logits(rig.ref.run(x))
    xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    np.testing.assert_array_equal(rig.runner.run_batch(xd, concurrency=0).cpu().numpy(), want)
    assert rig.runner.poll_error() is None
    # block 8 * k + i is member k of image i (conv_bgroup.hip): block 17 = member 2 of image 1 -- in EVERY group launch of the step
    set_opts(monkeypatch, bgroup_withhold=18, bgroup_polls=20000)
    rig.net.reload_options()
    assert any("conv_bgroup" in r["kernel"] and r["layer"] == withhold_row for r in rig.net.describe_launches(9, 0))
    got = rig.runner.run_batch(xd, concurrency=0)           # returns TF2_OK: the failure is on the device
    with pytest.raises(Tf2Error) as ei:
        rig.runner.poll_error()
    assert ei.value.status == -6 and "group launch" in str(ei.value)
    bad = got.cpu().numpy()
    assert (bad[1] != want[1]).any()                        # (image 1 went through a launch without one of its members)
    assert rig.runner.poll_error() is None                  # reported once, cleared
    # two steps without a poll in between: the report survives the next step's control-word initialisation
    rig.runner.run_batch(xd, concurrency=0)
    set_opts(monkeypatch, bgroup_withhold=None, bgroup_polls=None)
    rig.net.reload_options()
    clean = rig.runner.run_batch(xd, concurrency=0).cpu().numpy()
    with pytest.raises(Tf2Error):
        rig.runner.poll_error()
    np.testing.assert_array_equal(clean, want)              # the context, the handle and the workspace are alive
    assert rig.runner.poll_error() is None
    np.testing.assert_array_equal(rig.runner.run_batch(xd, concurrency=0).cpu().numpy(), want)


@pytest.mark.parametrize("pack_switch", ["nofast", "nodbl", "nosemi"])
def test_group_launches_with_other_packed_forms(r50, monkeypatch, pack_switch):
    """The group kernels share requant_epilogue.h with everything else: the wrap-exact generic requantisation on every row
    (nofast), plain instead of doubled channels (nodbl: more two-window rows -- some bottlenecks then fall back to
    separate launches, by the library's own eligibility rules), no SEMI rows.  Every layer against the oracle, group launches on."""
    set_opts(monkeypatch, **{pack_switch: "1"})
    set_opts(monkeypatch, bgroup_min7=1, bgroup_min14=1, bgroup_min28=1, bgroup_min56f=1, bfirst="1")
    set_opts(monkeypatch, alt_conc="0")
    rig = Rig(*r50, 0)
    groups = [r["kernel"] for r in rig.net.describe_launches(3, 0) if "conv_bgroup" in r["kernel"]]
    assert len(groups) >= 2 and sum("bottlenecks" in k for k in groups) >= 2      # (consecutive identity bottlenecks share a launch)
    x = synth.synth_images(rig.t, 3, 81, kind="int8")
    x[0, :, :3, :] = -128                                  # the negate quirk of pe.cl:32-37 on the way in
    rig.check_all_layers(x)
    x32 = synth.synth_images(rig.t, 32, 82)
    np.testing.assert_array_equal(rig.run(x32, keep_all=False)[:2], rig.ref.logits(rig.ref.run(x32[:2])))




@pytest.mark.parametrize("form", ["in_flight", "alone", "single_window", "generic"])
def test_first_bottleneck_as_one_launch_of_row_bands(r50, monkeypatch, form):
    """conv_bfirst.hip (round 6): rows 1-4 -- projection shortcut | reduce, 3x3, expand + residual from the shortcut, on the 56 x 56 maps -- as
    ONE launch of independent 4-row bands at two blocks per CU: the band's input stays in LDS for the reduce (halo rows recomputed) and the
    shortcut, the shortcut tile is the residual of the expand's epilogue, none of the three inner maps exists (keep_all: all written, so every
    layer can be read back).  The default with batches in flight; bfirst=2 one batch at a time as well (instead of the group launch).  Two-window
    rows (the shipped Q file) and one-window rows (a Q file without per-channel spread: the other instantiation), FAST and generic
    requantisation.  Every layer against the oracle at batch 2 and 5, batch-33 logits of repeated runs on the liveness-planned workspace,
    and against the plain launches."""
    alone = form == "alone"
    set_opts(monkeypatch, bfirst="2" if alone else "1", bfirst_min="1", alt_conc="0" if alone else "1")
    if form == "generic":
        set_opts(monkeypatch, nofast="1")
    t, q, model = r50
    if form == "single_window":
        q = synth.synth_q_values(t, 5, spread=0)
        model = synth.synth_model(t, q, 0)
    rig = Rig(t, q, model, 0)
    launches = rig.net.describe_launches(32, 0 if alone else 1)
    mine = [r for r in launches if "conv_bfirst" in r["kernel"]]
    assert [r["layer"] for r in mine] == [1] and mine[0]["grid"] == 32 * 14 and mine[0]["lds_bytes"] <= 80 * 1024
    assert ("dual" in mine[0]["kernel"]) == (form != "single_window")
    assert not any(r["layer"] in (2, 3, 4) for r in launches)
    x2 = synth.synth_images(rig.t, 2, 95, kind="int8")
    x2[0, :, :5, :] = -128
    rig.check_all_layers(x2)
    rig.check_all_layers(synth.synth_images(rig.t, 5, 96))
    x = synth.synth_images(rig.t, 33, 97)
    first = rig.run(x, keep_all=False).copy()
    np.testing.assert_array_equal(first[[0, 17, 32]], rig.ref.logits(rig.ref.run(x[[0, 17, 32]])))
    for _ in range(5):
        np.testing.assert_array_equal(rig.run(x, keep_all=False), first)
    set_opts(monkeypatch, bfirst="0")
    plain = Rig(t, q, model, 0)
    assert not any("conv_bfirst" in r["kernel"] for r in plain.net.describe_launches(32, 0 if alone else 1))
    np.testing.assert_array_equal(plain.run(x, keep_all=False), first)


@pytest.mark.parametrize("form", ["in_flight", "alone", "single_window", "generic", "split_k_rows"])
def test_short_k_pointwise_rows_with_the_k_extent_in_lds(r50, monkeypatch, form):
    """conv_pwk.hip (round 6): 1x1 rows of 128, 256 or 512 input channels (ResNet-50's 256 -> 64, 256 -> 128 | 512 / 2, 128 -> 512, 512 -> 256 | 1024 / 2,
    256 -> 1024, 512 -> 2048)
    with the block's pixels resident in LDS (fetched once for every output channel) and a wave's weight fragments resident in registers per
    pass of 32 channels; two-window rows are swept window by window into one accumulator set.  The default with batches
    in flight (rows of >= 4096 pixels; pwk=0: the ring kernel), pwk=2 one batch at a time as well.  Here every eligible row (pwk_minpix=0, pwk_units=0: rows of any size; pwk_slabs=8: the 512-channel rows too, which the default leaves to the ring kernel; split_k_rows: the rows the
    in-block split-K kernel would take as well), stride 1 and 2, with and without residual, one- and two-window packing, FAST and generic
    requantisation, ragged pixel counts (batch 2 / 5: tiles that straddle the end), one to eight channel parts per pixel tile, one to three tiles per block;
    every layer against the oracle, batch-33 logits of repeated runs on the liveness-planned workspace, and against the plain launches."""
    alone = form == "alone"
    set_opts(monkeypatch, pwk="2" if alone else "1", pwk_minpix="0", pwk_units="0", pwk_slabs="8", alt_conc="0" if alone else "1")
    if form == "split_k_rows":
        set_opts(monkeypatch, pwk_sk="1")
    if form == "generic":
        set_opts(monkeypatch, nofast="1")
    t, q, model = r50
    if form == "single_window":
        q = synth.synth_q_values(t, 5, spread=0)
        model = synth.synth_model(t, q, 0)
    rig = Rig(t, q, model, 0)
    conc = 0 if alone else 1
    mine = {r["layer"]: r["kernel"] for r in rig.net.describe_launches(33, conc) if "conv_pwk" in r["kernel"]}
    assert {5, 8, 11, 14, 24, 27} <= set(mine) and "conv_pwk_pair_kernel" in mine[11] and "conv_pwk_pair_kernel<8 slabs" in mine[24] and 12 not in mine, mine
    if form == "split_k_rows":
        assert {30, 33} <= {r["layer"] for r in rig.net.describe_launches(2, conc) if "conv_pwk" in r["kernel"]}
    assert any("x 1 channel parts" in k for k in mine.values()) and any("x 4 channel parts" in k for k in mine.values())
    set_opts(monkeypatch, pair="0" if form == "generic" else None)       # (one form with rows 11 and 12 as launches of their own)
    if form == "generic":
        rig.net.reload_options()
        assert {11, 12} <= {r["layer"] for r in rig.net.describe_launches(33, conc) if "conv_pwk_kernel" in r["kernel"]}
    if form != "single_window":
        assert any("dual" in k for k in mine.values()) and any("single" in k for k in mine.values())
    x2 = synth.synth_images(rig.t, 2, 101, kind="int8")
    x2[0, :, :5, :] = -128
    rig.check_all_layers(x2)
    rig.check_all_layers(synth.synth_images(rig.t, 5, 102))
    x = synth.synth_images(rig.t, 33, 103)
    first = rig.run(x, keep_all=False).copy()
    np.testing.assert_array_equal(first[[0, 17, 32]], rig.ref.logits(rig.ref.run(x[[0, 17, 32]])))
    for _ in range(5):
        np.testing.assert_array_equal(rig.run(x, keep_all=False), first)
    set_opts(monkeypatch, pwk="0")
    plain = Rig(t, q, model, 0)
    assert not any("conv_pwk" in r["kernel"] for r in plain.net.describe_launches(33, conc))
    np.testing.assert_array_equal(plain.run(x, keep_all=False), first)


@pytest.mark.parametrize("rows,conc", [("7", "1"), ("4", "1"), ("2", "0")])
def test_band_launches_of_the_identity_bottlenecks(r50, monkeypatch, rows, conc):
    """conv_bband.hip: the identity bottlenecks of stage 3 (rows 15-23: two-window reduce, the last one's 3x3 two-window too) and stage 4
    (rows 28-42) as ONE launch each of independent row bands -- a block owns
    `rows` output rows of one image and all channels, recomputes the reduce for its halo rows, keeps both intermediates in LDS; no
    exchange between blocks.  The default with batches in flight (7 rows), bband=2 one batch at a time as well.  Every
    layer against the oracle at batch 2 and 5 (keep_all: the intermediates are written out too), then batch-32 logits of repeated
    runs on the liveness-planned workspace, and against the plain launches."""
    set_opts(monkeypatch, bband="2" if conc == "0" else "1")
    set_opts(monkeypatch, bband_rows=rows)
    set_opts(monkeypatch, bband_rows_alone=rows)
    set_opts(monkeypatch, bband_min="1")
    set_opts(monkeypatch, alt_conc=conc)
    rig = Rig(*r50, 0)
    launches = rig.net.describe_launches(32, int(conc))
    stage3 = [15, 18, 21] if rows != "2" else []          # (28 x 28: bands of 7 or 4 rows)
    assert [r["layer"] for r in launches if "conv_bband" in r["kernel"]] == stage3 + [28, 31, 34, 37, 40]
    assert not any(r["layer"] in (29, 30, 32, 33) for r in launches)
    if stage3:
        k = {r["layer"]: r["kernel"] for r in launches}
        assert "dual reduce" in k[15] and "dual 3x3" not in k[15] and "dual reduce,dual 3x3" in k[21]
    rig.check_all_layers(synth.synth_images(rig.t, 2, 91))
    rig.check_all_layers(synth.synth_images(rig.t, 5, 92))
    x = synth.synth_images(rig.t, 32, 93)
    first = rig.run(x, keep_all=False).copy()
    np.testing.assert_array_equal(first[:3], rig.ref.logits(rig.ref.run(x[:3])))
    for _ in range(5):
        np.testing.assert_array_equal(rig.run(x, keep_all=False), first)
    set_opts(monkeypatch, bband="0")
    plain = Rig(*r50, 0)
    assert not any("conv_bband" in r["kernel"] for r in plain.net.describe_launches(32, int(conc)))
    np.testing.assert_array_equal(plain.run(x, keep_all=False), first)


@pytest.mark.parametrize("merge", ["fire", "1", "0"])
def test_merged_expand_rows(merge, monkeypatch):
    """PackLayer::merge_next (weight_pack.cpp, round 5): a fire module's expand1x1 and expand3x3 rows -- same input tensor, adjacent
    slices of one concat tensor (kNStart / kBranchTail, quantization.cpp:42-49) -- run as ONE 3x3 launch whose first rows carry the 1x1
    filters as centre taps (and, for fire3 / fire5, ONE pool launch).  Every row of a 67 x 67 and a 131 x 131 SqueezeNet 1.1 against the
    oracle with the rows merged and separate (doubled squeeze outputs, 64- and 128-row tiles, pooled and unpooled pairs)."""
    # "fire" (the default): the merged rows AND, for unpooled fire modules on 56 / 28 / 14-wide maps, the squeeze in the same launch
    # (conv_fire.hip: squeeze over the halo rows, its requantised output in LDS planes, the merged expand from there) -- 227 x 227 images
    set_opts(monkeypatch, merge="0" if merge == "0" else "1", fire="1" if merge == "fire" else "0")
    for hw, seed, b, spread in ((67, 6, 5, 2), (131, 8, 3, 2)) + (((227, 6, 2, 1), (227, 10, 3, 2)) if merge == "fire" else ()):
        t = cfg.squeezenet11_tables(image_hw=hw)
        q = synth.synth_q_values(t, seed, spread=spread)
        rig = Rig(t, q, synth.synth_model(t, q, seed), 0)
        ls = rig.net.describe_launches(b, 1)
        rows = {r["layer"] for r in ls}
        assert (3 in rows) == (merge == "0") and (6 in rows) == (merge == "0")
        if hw == 227 and seed == 6:                        # (bench.py's Q values: every squeeze output doubled, the merged expands one-window)
            assert sum("conv_fire" in r["kernel"] for r in ls) >= 2
        rig.check_all_layers(synth.synth_images(t, b, seed))


@pytest.mark.parametrize("first", ["1", "0"])
def test_first_layer_with_its_input_preparation_in_one_launch(first, monkeypatch):
    """conv_first_kernel (round 5): a 3x3 / stride 1 first layer on the 3-channel image -- quantisation, the im2col tile (kept in LDS) and
    the pointwise MFMA layer over it in ONE launch -- against the oracle: the quantised image read back (-128 pixels included), row 0 and
    everything behind it; float and int8 inputs, one- and two-window first layers (spread 1 / 2), ragged widths (40, 72: several rows per
    block; 176: one row), batch 1 and 5; first=0: the two separate launches."""
    set_opts(monkeypatch, first=first)
    for hw, seed, spread, kind, b in ((40, 3, 1, "int8", 5), (72, 4, 2, "float", 2), (176, 5, 2, "float", 1)):
        t = cfg.vgg16_tables(hw, 10) if hw >= 64 else cfg.vgg16_tables(hw, 10, with_fc=False)
        q = synth.synth_q_values(t, seed, spread=spread)
        rig = Rig(t, q, synth.synth_model(t, q, seed), 0)
        k0 = rig.net.describe_launches(b, 1)[0]
        assert ("conv_first_kernel" in k0["kernel"] and k0["layer"] == 0) == (first == "1"), k0
        x = synth.synth_images(t, b, seed, kind=kind)
        if kind == "int8":
            x[0, :, :3, :] = -128
        rig.check_all_layers(x, layers={0, 1, 2})


@pytest.mark.parametrize("first_pool", ["1", "0"])
def test_first_layer_with_its_pool_in_one_launch(first_pool, monkeypatch):
    """conv_first_pool_kernel (round 5): SqueezeNet 1.1's front -- input quantisation, the stride-2 3x3 conv1 over the im2col tile and the
    3x3 / 2 ceil-mode pool1, conv map in LDS -- in ONE launch, against the oracle: the quantised image read back, the conv map, the pooled
    tensor and the rows behind it; float and int8 inputs (-128 pixels included), image sizes whose last pool window hangs over the map
    (ceil mode: 47 -> 23 -> 11 wide with one column / row beyond it) and not (227), one- and two-window conv1 (spread 1 / 2), batch 1 and 3;
    first_pool=0: the three separate launches."""
    set_opts(monkeypatch, first_pool=first_pool)
    for hw, seed, spread, kind, b in ((47, 3, 1, "int8", 3), (63, 4, 2, "float", 2), (227, 6, 1, "float", 1)):
        t = cfg.squeezenet11_tables(hw)
        q = synth.synth_q_values(t, seed, spread=spread)
        rig = Rig(t, q, synth.synth_model(t, q, seed), 0)
        k0 = rig.net.describe_launches(b, 1)[0]
        assert ("conv_first_pool_kernel" in k0["kernel"] and k0["layer"] == 0) == (first_pool == "1"), k0
        x = synth.synth_images(t, b, seed, kind=kind)
        if kind == "int8":
            x[0, :, :3, :] = -128
        want = rig.check_all_layers(x, layers={0, 1, 2, 3})
        np.testing.assert_array_equal(rig.run(x, keep_all=False), want)         # ... and the plan of a plain run (no per-layer tensors kept)
