"""The C-ABI library on the CPU: every symbol include/tf2_amd.h declares is exported, the
host-side (load-time) entry points agree with the oracle on networks the reference ships no
header for, and errors are status codes, never exit()."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import netref
from tf2_amd import _lib, config as cfg, network, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests.conftest import set_opts  # noqa: E402


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tf2_amd.h")).read()
    declared = set(re.findall(r"\b(tf2_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tf2_status"}
    L = C.CDLL(_lib.LIB_PATH)
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().tf2_abi_version() == 1 and _lib.lib().tf2_has_device_code() == 1


@pytest.mark.parametrize("which", ["tiny", "squeezenet", "vgg_small"])
def test_host_numerics_match_oracle(which):
    t = {"tiny": cfg.tiny_tables, "squeezenet": cfg.squeezenet11_tables,
         "vgg_small": lambda: cfg.vgg16_tables(32, 10)}[which]()
    q = synth.synth_q_values(t, seed=3)
    model = synth.synth_model(t, q, seed=3)
    net = network.NetWork(t)
    qt = net.Quantization(synth.q_text(q))
    net.LoadModel(model)
    R = netref.RefNet(t, q, model)
    np.testing.assert_array_equal(qt, R.q)
    for L in R.plan:
        np.testing.assert_array_equal(net.codes(L.index), R.codes[L.index])
        for a, b in zip(net.bias_bn(L.index), R.bn[L.index]):
            np.testing.assert_array_equal(a, b)
    net.Pack(0)
    blob = net.packed_host()
    assert blob.size > 0 and net.workspace_size(4) > 0 and net.workspace_size(4, True) >= net.workspace_size(4)
    # a second handle with the same tables adopts the image (what non-root ranks do)
    other = network.NetWork(t)
    other.Quantization(synth.q_text(q))
    other.adopt_packed(blob)
    np.testing.assert_array_equal(other.packed_host(), blob)
    # ... and one with different tables refuses it
    t2 = cfg.tiny_tables(hw=16)
    stranger = network.NetWork(t2)
    with pytest.raises(_lib.Tf2Error):
        stranger.adopt_packed(blob)


def test_errors_are_status_codes():
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 0)
    model = synth.synth_model(t, q, 0)
    net = network.NetWork(t)
    with pytest.raises(_lib.Tf2Error) as e:
        net.LoadModel(model)                       # q table not set yet
    assert e.value.status == -2
    net.Quantization(synth.q_text(q))
    with pytest.raises(_lib.Tf2Error) as e:
        net.LoadModel(model[:-5])                  # truncated stream
    assert e.value.status == -3 and "short" in str(e.value)
    with pytest.raises(_lib.Tf2Error) as e:
        net.LoadModel(np.concatenate([model, model[:3]]))   # trailing floats
    assert e.value.status == -3
    with pytest.raises(_lib.Tf2Error):
        net.Pack(0)                                # no model loaded
    net.LoadModel(model)
    with pytest.raises(_lib.Tf2Error):
        net.Pack(7)
    # inconsistent tables are rejected at create time
    bad = cfg.tiny_tables()
    bad["kInputChannels"][2] = 99
    with pytest.raises(_lib.Tf2Error):
        network.NetWork(bad)
    # short Q files read as zeros like fscanf on EOF (quantization.cpp:45)
    net2 = network.NetWork(t)
    net2.Quantization(b"1 2 3\n")
    assert net2.q_values_read == cfg.q_value_count(t) and (net2.q[1:] == 0).all() and net2.q[0, 2] == -3


def test_topk_argument_checks():
    lg = np.zeros(10, np.int8); q = np.zeros(10, np.int8)
    lab = np.zeros(5, np.int32)
    st = _lib.lib().tf2_topk(lg.ctypes.data, q.ctypes.data, 10, 11, lab.ctypes.data, None)
    assert st == -1
    q[3] = 1            # Q < 0 for the last layer: the reference shifts by a negative count (UB); we refuse
    st = _lib.lib().tf2_topk(lg.ctypes.data, q.ctypes.data, 10, 5, lab.ctypes.data, None)
    assert st == -1


def test_pack_modes_and_phase_structure():
    """mode 0: MFMA everywhere an int8 tile decomposition applies; 1: shift kernel for k>1;
    2: shift kernel everywhere.  All three pack without error and differ in size."""
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 1)
    model = synth.synth_model(t, q, 1)
    sizes = []
    for mode in (0, 1, 2):
        net = network.NetWork(t)
        net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(mode)
        sizes.append(net.packed_host().size)
    assert len(set(sizes)) == 3


def test_logits_size_and_third_party_table_checks():
    """tf2_net_logits_size = the dense output [batch][H_last * W_last][N_last]; descs that would index outside the q table
    or a channel row (a hand-edited <net>.h, a third-party caller) are refused at create time with the layer named."""
    t = cfg.tiny_tables()
    net = network.NetWork(t)
    N = net.plan[-1].N
    assert _lib.lib().tf2_net_logits_size(net._h, 5) == 5 * N
    v = cfg.vgg16_tables(64, 10, with_fc=False)           # ends on a 2x2 feature map, not on a 1x1 classifier
    nv = network.NetWork(v)
    L = nv.plan[-1]
    assert _lib.lib().tf2_net_logits_size(nv._h, 3) == 3 * L.PH * L.PW * L.N and L.PH * L.PW > 1
    lib = _lib.lib()

    def create(mut):
        plan, nd, arr = network._layer_descs(t)
        mut(nd, arr)
        h = C.c_void_p()
        st = lib.tf2_net_create(C.byref(nd), arr, C.byref(h))
        msg = lib.tf2_last_error().decode()
        if st == 0:
            lib.tf2_net_destroy(h)
        return st, msg

    st, msg = create(lambda nd, a: setattr(a[1], "q_in_row", 99))
    assert st == -1 and "layer 1" in msg and "q_in_row" in msg
    st, msg = create(lambda nd, a: setattr(a[2], "n_start", nd.max_out_channel))
    assert st == -1 and "layer 2" in msg
    st, msg = create(lambda nd, a: setattr(nd, "max_out_channel", 4))
    assert st == -1
    st, msg = create(lambda nd, a: None)
    assert st == 0
    # options are re-read on request (kernel A/B switches for tests and tools); harmless on the CPU
    assert lib.tf2_net_reload_options(net._h) == 0


def test_describe_launches_is_the_librarys_own_selection(golden_dir):
    """tf2_net_describe_launches: the launch list of a step as Net::launch_plan and the launchers select it -- what
    tools/pmc_summary.py uses to attach rocprofv3 rows to layers.  No device needed."""
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(synth.synth_model(t, q, 0)); net.Pack(0)
    one = net.describe_launches(32, 0)
    many = net.describe_launches(32, 1)
    assert one[0]["layer"] == -1 and "prep" in one[0]["kernel"]
    assert one[1]["layer"] == 0 and "conv_stem_pool" in one[1]["kernel"]        # conv1 + its 3x3/2 max pool: one launch
    assert not any(r["kernel"] == "maxpool_kernel" for r in one)
    assert [r["layer"] for r in one] == sorted(r["layer"] for r in one)
    # fused pairs (conv_bneck) carry their first row; with batches in flight the 128-channel pairs run unfused
    fused_one = {r["layer"] for r in one if "conv_bneck" in r["kernel"]}
    fused_many = {r["layer"] for r in many if "conv_bneck" in r["kernel"]}
    assert fused_one == {6, 9} and fused_many == {6, 9}             # (the 128-channel pairs belong to group launches one batch at a time; rows 3-4 to conv_bfirst)
    # round 6: rows 1-4 (projection shortcut | reduce, 3x3, expand + residual on the 56 x 56 maps) are ONE launch of independent 4-row bands
    # at two blocks per CU in BOTH plans (conv_bfirst.hip: 14 bands per image, under 80 KB of LDS; bfirst=1 keeps the group launch of the
    # one-batch plan, bfirst=0 the separate launches)
    for plan_rows in (one, many):
        bf = [r for r in plan_rows if "conv_bfirst" in r["kernel"]]
        assert [r["layer"] for r in bf] == [1] and bf[0]["grid"] == 32 * 14 and bf[0]["block"] == 512 and 2 * bf[0]["lds_bytes"] <= 160 * 1024
        assert "dual" in bf[0]["kernel"] and not any(r["layer"] in (2, 3, 4) for r in plan_rows)
    # independent neighbouring rows in one launch: the shortcut convolution of stages 3 and 4 (and, with the wide tiles of the
    # several-streams plan, stage 5) next to the first 1x1 of the stage's first bottleneck
    assert {r["layer"] for r in one if "pair" in r["kernel"]} == {11, 24} and {r["layer"] for r in many if "pair" in r["kernel"]} == {11, 24, 43}
    assert {r["layer"] for r in many if "conv_pwk_pair" in r["kernel"]} == {11}              # (round 6: the in-flight plan's 256-channel pair on conv_pwk.hip)
    # one batch at a time the identity bottlenecks of stages 3 and 5 are ONE launch each (conv_bgroup.hip: rows 15-17 ..., 47-49, 50-52),
    # the five of stage 4 (rows 28-42) ONE launch together
    groups = [r for r in one if "conv_bgroup" in r["kernel"]]
    assert [r["layer"] for r in groups] == [15, 21, 28, 47] and all(r["grid"] == 256 and r["block"] == 512 for r in groups)
    assert "x 2 bottlenecks" in [r for r in groups if r["layer"] == 15][0]["kernel"] and "dual 3x3" in [r for r in groups if r["layer"] == 21][0]["kernel"]
    assert "x 5 bottlenecks" in [r for r in groups if r["layer"] == 28][0]["kernel"]
    assert "x 2 bottlenecks" in [r for r in groups if r["layer"] == 47][0]["kernel"] and "global average" in [r for r in groups if r["layer"] == 47][0]["kernel"]
    assert not any("conv_bgroup" in r["kernel"] for r in many)
    # with batches in flight the identity bottlenecks of stages 3 and 4 are ONE band launch each (conv_bband.hip: no exchange between
    # blocks -- two-window reduce on stage 3, the last one's 3x3 two-window as well; 7-row bands everywhere since round 6), stage 5 keeps its separate launches
    bands = [r for r in many if "conv_bband" in r["kernel"]]
    assert [r["layer"] for r in bands] == [15, 18, 21, 28, 31, 34, 37, 40] and not any("conv_bband" in r["kernel"] for r in one)
    assert [r["grid"] for r in bands] == [128, 128, 128, 64, 64, 64, 64, 64] and all(r["block"] == 512 and r["lds_bytes"] + 65536 <= 160 * 1024 for r in bands)
    assert "dual reduce" in bands[0]["kernel"] and "dual reduce,dual 3x3" in bands[2]["kernel"] and "dual" not in bands[3]["kernel"]
    # the global average: inside the last expand's split-K launch one batch at a time; with batches in flight the expand runs on its
    # 208-block 128 x 128 tiles and the average is a launch of its own (round 6: the 1024-block fused launch cost 13 us of the in-flight step
    # against 2.6), and the split-K launches take two ring stages there (64 KiB of LDS: another block fits beside them)
    assert "global average" in [r for r in one if r["layer"] == 47][0]["kernel"] and not any(r["kernel"] == "global_avg_kernel" for r in one)
    assert [r["kernel"].split("<")[0] for r in many if r["layer"] == 52] == ["conv_mfma2_kernel", "global_avg_kernel"]
    assert all("S2" in r["kernel"] for r in many if "conv_mfma_sk" in r["kernel"])
    assert len(many) == 32 and len(one) == 22
    assert [r["grid"] for r in net.describe_launches(40, 0) if r["layer"] == 28] == [256, 64]     # at most 32 images per launch
    # every ring-kernel launch of ResNet-50 takes the arithmetic-gather instantiation (single-window and dual layers are dense)
    ring = [r for r in one if "conv_mfma" in r["kernel"]]
    assert ring and all("dense" in r["kernel"] and "tables" not in r["kernel"] for r in ring)
    assert all(0 < r["grid"] and r["block"] in (256, 512) and 0 <= r["lds_bytes"] <= 160 * 1024 for r in one)
    # batch 1: small grids, the split-K kernel on the 64-row layers; the shortcut convolution of stages 4 / 5 and the first 1x1 of the stage's first
    # bottleneck -- independent rows, the same split-K instantiation -- share a launch (round 6: 53 launches instead of 55)
    b1 = net.describe_launches(1, 0)
    assert any("conv_mfma_sk" in r["kernel"] for r in b1)
    assert [r["layer"] for r in b1 if "conv_mfma_sk_pair_kernel" in r["kernel"]] == [24, 43] and not any(r["layer"] in (25, 44) for r in b1)
    # ... and so do rows 1 | 2 and 11 | 12 on the ring kernel's 64-row four-wave shape (round 6: row 1 has a 64-row alternative for these grids): 51 launches, 340 us
    small_pairs = [r for r in b1 if "conv_mfma2_pair_kernel<2x2 waves of 32x32" in r["kernel"]]
    assert [r["layer"] for r in small_pairs] == [1, 11] and small_pairs[0]["grid"] == 196 + 49 and small_pairs[0]["block"] == 256 and len(b1) == 51
    assert not any(r["layer"] in (2, 12) for r in b1)
    # ... and the 8-block split-K launches of the 7 x 7 maps split K over blocks as well (round 6: 64 blocks, the last ticket of a tile finishes it)
    kb = {r["layer"]: r for r in b1 if "K over 8 blocks" in r["kernel"]}
    assert sorted(kb) == [45, 47, 48, 50, 51] and all(r["grid"] == 64 for r in kb.values())
    assert not any("K over" in r["kernel"] for r in net.describe_launches(2, 0)) and not any("K over" in r["kernel"] for r in one + many)
    with pytest.raises(_lib.Tf2Error):
        net.describe_launches(0, 0)


def test_short_k_pointwise_option_changes_exactly_its_rows(golden_dir, monkeypatch):
    """pwk (conv_pwk.hip, default 1: with batches in flight): the in-flight plan sends ResNet-50's dense 1x1 rows of 128 / 256 input channels and
    enough (tile, channel part) units that a block walks several tiles -- 5, 8, 11 | 12 (one launch of two rows, like the ring kernel's pair), 14 -- to the
    kernel that keeps a block's weight fragments in registers and streams pixel tiles through two LDS buffers; pwk_units=0 adds the rows of fewer tiles
    (27: its wide-tile alternative shares the main entry's weight tiles, so it is taken on the main entry), pwk_slabs=8 the 512-channel rows (24 | 25) --
    both measured slower than the ring kernel, not the default.  pwk=0 gives the ring-kernel plan back, nothing else
    moves; the one-batch plan takes it only with pwk=2.  No device needed."""
    from tests.conftest import set_opts
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    model = synth.synth_model(t, q, 0)

    def plans():
        net = network.NetWork(t)
        net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
        return net.describe_launches(32, 1), net.describe_launches(32, 0)
    many1, one1 = plans()
    mine = {r["layer"]: r for r in many1 if "conv_pwk" in r["kernel"]}
    assert sorted(mine) == [5, 8, 11, 14] and not any("conv_pwk" in r["kernel"] for r in one1)
    assert "conv_pwk_pair_kernel" in mine[11]["kernel"] and not any(r["layer"] == 12 for r in many1)
    assert all(r["block"] == 256 and 190 <= r["grid"] <= 260 for r in mine.values()), mine       # about one block per CU, 3-7 tiles each
    assert "4 slabs,2 channel groups,dual" in mine[5]["kernel"] and "x 1 channel parts" in mine[5]["kernel"] and "of 4..4 tiles" in mine[5]["kernel"]
    assert "2 slabs,4 channel groups,single" in mine[14]["kernel"] and "x 4 channel parts" in mine[14]["kernel"]
    set_opts(monkeypatch, pwk="0")
    many0, one0 = plans()
    assert not any("conv_pwk" in r["kernel"] for r in many0 + one0)
    assert [(r["layer"], r["kernel"]) for r in one1] == [(r["layer"], r["kernel"]) for r in one0]
    rows = (5, 8, 11, 14)
    others0 = [(r["layer"], r["kernel"], r["grid"]) for r in many0 if r["layer"] not in rows]
    others1 = [(r["layer"], r["kernel"], r["grid"]) for r in many1 if r["layer"] not in rows]
    assert others0 == others1 and len(many1) == len(many0)
    set_opts(monkeypatch, pwk="1", pwk_slabs="8", pwk_units="0")
    many8, _ = plans()
    k8 = {r["layer"]: r["kernel"] for r in many8 if "conv_pwk" in r["kernel"]}
    assert sorted(k8) == [5, 8, 11, 14, 24, 27] and "conv_pwk_pair_kernel<8 slabs" in k8[24] and not any(r["layer"] == 25 for r in many8)
    assert "x 8 channel parts" in k8[27]
    set_opts(monkeypatch, pwk="2", pwk_slabs=None, pwk_units=None)
    _, one2 = plans()
    assert {5, 8} <= {r["layer"] for r in one2 if "conv_pwk" in r["kernel"]}


def test_launch_selection_of_the_vgg_style_networks(monkeypatch):
    """The other BASELINE configurations' plans, without a device: a 3x3 first layer on the 3-channel image runs as a pointwise layer
    over the im2col image the input kernel writes (Net::init); the 3x3 / 1 / pad 1 body layers of maps >= 14 x 14 take conv_c3.hip
    (64 / 128 / 256 output channels per block by channel count, windows and grid); both can be switched off per handle."""
    def launches(t, batch, seed=0):
        q = synth.synth_q_values(t, seed, spread=1)
        net = network.NetWork(t)
        net.Quantization(synth.q_text(q)); net.LoadModel(synth.synth_model(t, q, seed)); net.Pack(0)
        return net.describe_launches(batch, 0)
    v = launches(cfg.vgg16_tables(), 32)
    # round 5: input preparation + conv1_1 in ONE launch (conv_first_kernel: the im2col tile stays in LDS), attributed to table row 0
    assert "conv_first_kernel<im2col" in v[0]["kernel"] and v[0]["layer"] == 0 and v[1]["layer"] == 1
    c3 = [r for r in v if "conv_c3" in r["kernel"]]
    assert [r["layer"] for r in c3] == list(range(1, 13))                      # conv1_2 .. conv5_3
    # conv1_2: its 2x2 pool rides in the launch (round 5: tiles of 8 x 32 pixels, a column tile of the MFMA layout = a tile row); 7168 tiles
    # walked by one block per CU, weights resident
    assert c3[0]["kernel"].startswith("conv_c3_w9_kernel<64 channels x 8x32 pixels") and "2x2 pool" in c3[0]["kernel"] and c3[0]["grid"] == 256
    assert {r["layer"] for r in c3 if "2x2 pool" in r["kernel"]} == {1, 3, 6, 9} and [r["layer"] for r in v if "maxpool" in r["kernel"]] == [12]
    assert all(r["grid"] <= 256 for r in c3 if "<128 channels" in r["kernel"])  # 128-channel blocks: at most what the chip holds at once
    assert all(r["block"] == 512 and r["lds_bytes"] <= 8192 for r in c3)
    sizes = {r["layer"]: int(r["kernel"].split("<")[1].split(" ")[0]) for r in c3}
    duals = {r["layer"] for r in c3 if "dual" in r["kernel"]}
    assert all(sizes[l] in (64, 128) for l in duals)                            # two accumulator sets: one row tile per wave
    assert all(sizes[l] in (64, 256) for l in sizes if l >= 5 and l not in duals)      # one-window layers of 256+ channels: 256 per block where the grid allows
    assert sizes[11] == 64 or 11 in duals                                        # 14 x 14: 32 tiles -- the small grid takes 64-channel blocks
    s = launches(cfg.squeezenet11_tables(), 32)
    # round 5: input preparation + the stride-2 conv1 + pool1 in ONE launch (conv_first_pool_kernel: im2col tile and conv map stay in LDS)
    assert s[0]["kernel"].startswith("conv_first_pool_kernel<stride 2,2 pooled rows") and s[0]["layer"] == 0 and s[0]["grid"] == 32 * 28 and s[1]["layer"] == 1
    assert s[0]["lds_bytes"] <= 78 * 1024 and s[-1]["kernel"] == "conv_shift_fc_kernel"
    # merged rows (round 5): every fire module's expand1x1 | expand3x3 pair is ONE launch (PackLayer::merge_next) -- 24 launches, not 34,
    # none on the second row of a pair; merge=0 brings the separate rows back (its 64-channel 3x3 rows on 14 x 14 then take conv_c3)
    # ... and a fire module (squeeze + merged expands) on a map >= 28 wide is ONE launch of row bands (conv_fire.hip), a 3x3 / 2 pool behind the
    # expands included (fire3, fire5) in the plan of one batch at a time: 16 launches (with batches in flight those two pools are launches of
    # their own, 18: the pooled blocks own their CU, measured slower there)
    # (fire_pool=2: the pools as launches in both plans; fire=1: the 14 x 14 modules as well, 12 -- measured slower; fire_pool=0: only the
    #  unpooled modules, 20; fire=0: 22; first_pool=0: the front as three launches, 24)
    assert len(s) == 16 and [r["layer"] for r in s if "conv_fire" in r["kernel"]] == [1, 4, 7, 10] and not any("maxpool" in r["kernel"] for r in s)
    assert [r["layer"] for r in s if "3x3/2 pool" in r["kernel"]] == [4, 10]
    set_opts(monkeypatch, fire_pool="2")
    s4 = launches(cfg.squeezenet11_tables(), 32)
    assert len(s4) == 18 and [r["layer"] for r in s4 if "maxpool" in r["kernel"]] == [5, 11]
    set_opts(monkeypatch, fire_pool=None)
    nq = synth.synth_q_values(cfg.squeezenet11_tables(), 0, spread=1)
    n5 = network.NetWork(cfg.squeezenet11_tables())
    n5.Quantization(synth.q_text(nq)); n5.LoadModel(synth.synth_model(cfg.squeezenet11_tables(), nq, 0)); n5.Pack(0)
    assert len(n5.describe_launches(32, 1)) == 18 and len(n5.describe_launches(32, 0)) == 16
    set_opts(monkeypatch, fire="1", fire_pool=None)
    s2 = launches(cfg.squeezenet11_tables(), 32)
    assert len(s2) == 12 and [r["layer"] for r in s2 if "conv_fire" in r["kernel"]] == [1, 4, 7, 10, 13, 16, 19, 22]
    set_opts(monkeypatch, fire="2", fire_pool="0")
    s3 = launches(cfg.squeezenet11_tables(), 32)
    assert len(s3) == 20 and [r["layer"] for r in s3 if "conv_fire" in r["kernel"]] == [1, 7]
    set_opts(monkeypatch, fire="0", first_pool="0", fire_pool=None)
    s1 = launches(cfg.squeezenet11_tables(), 32)
    assert "im2col" in s1[0]["kernel"] and s1[0]["layer"] == -1 and "conv_pw" in s1[1]["kernel"] and "maxpool" in s1[2]["kernel"]      # (stride 2: its own input kernel)
    assert len(s1) == 24 and not {3, 6, 9, 12, 15, 18, 21, 24} & {r["layer"] for r in s1}
    set_opts(monkeypatch, merge="0")
    s0 = launches(cfg.squeezenet11_tables(), 32)
    set_opts(monkeypatch, merge=None, fire=None, first_pool=None)
    assert len(s0) == 34 and {r["layer"] for r in s0 if "conv_c3" in r["kernel"]} == {21, 24}
    assert any("conv_c3" in r["kernel"] for r in launches(cfg.ssd300_tables(), 32))
    set_opts(monkeypatch, c3="0"); set_opts(monkeypatch, im2col0="0")
    v0 = launches(cfg.vgg16_tables(), 32)
    assert not any("conv_c3" in r["kernel"] or "im2col" in r["kernel"] for r in v0) and "conv_mfma2" in v0[1]["kernel"]


def test_run_ex_rejects_bad_options_without_touching_the_device():
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 0)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(synth.synth_model(t, q, 0)); net.Pack(0)
    L = _lib.lib()
    buf = (C.c_char * 64)()
    assert C.sizeof(_lib.RunOpts) == 24
    for o in (_lib.RunOpts(8, 0, -1, -1, None),                      # older / truncated struct
              _lib.RunOpts(24, 0, 2, -1, None),                      # concurrency out of range
              _lib.RunOpts(24, 0, 0, 999, C.cast(buf, C.c_void_p))):  # mark layer outside the table
        assert L.tf2_net_run_ex(net._h, buf, 1, buf, 64, buf, None, C.byref(o)) == -1, L.tf2_last_error()
    assert L.tf2_net_run_ex(net._h, buf, 1, buf, 64, buf, None, None) == -1


def _liveness_net(golden_dir, name):
    if name == "resnet50":
        t = cfg.resnet50_tables()
        q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    elif name == "googlenet":                               # concat slices and independent pooling rows
        import json
        t = cfg.NetTables(json.load(open(os.path.join(golden_dir, "tables_googlenet.json"))))
        t.setdefault("xConv1Rewrite", 1)
        q = np.loadtxt(os.path.join(golden_dir, "googlenet_Q"), dtype=np.int32)
    else:
        t = cfg.squeezenet11_tables()
        q = synth.synth_q_values(t, 21, spread=2)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(synth.synth_model(t, q, 0)); net.Pack(0)
    return net


@pytest.mark.parametrize("name,batch", [("resnet50", 5), ("resnet50", 32), ("resnet50", 40), ("googlenet", 8), ("squeezenet", 32)])
def test_rows_that_share_a_launch_never_share_memory(golden_dir, name, batch):
    """tf2_net_describe_workspace: (1) the first-fit planner never gives two tensors that are alive at the same table row overlapping
    bytes; (2) every launch that computes several rows -- conv_bneck pairs, pair launches, group launches and their chains of up
    to five bottlenecks, whose blocks are ordered by flags and cache scopes instead of kernel boundaries -- touches pairwise
    disjoint memory, and everything it touches is alive for the whole launch (Net::plan).  One batch at a time and the
    several-streams plan; no device needed."""
    net = _liveness_net(golden_dir, name)
    tensors, rows = net.describe_workspace(batch)
    nl = len(rows)
    assert nl == len(net.plan) and all(0 <= x["offset"] and 0 < x["bytes"] and x["first_row"] <= x["last_row"] for x in tensors)
    assert max(x["offset"] + x["bytes"] for x in tensors) <= net.workspace_size(batch)

    def overlap(a, b):
        return a["offset"] < b["offset"] + b["bytes"] and b["offset"] < a["offset"] + a["bytes"]

    for i, a in enumerate(tensors):
        for b in tensors[i + 1:]:
            if a["first_row"] <= b["last_row"] and b["first_row"] <= a["last_row"]:
                assert not overlap(a, b), (a, b)
    n_multi = 0
    for conc in (0, 1):
        starts = sorted({r["layer"] for r in net.describe_launches(batch, conc) if r["layer"] >= 0})
        for k, l in enumerate(starts):
            end = (starts[k + 1] if k + 1 < len(starts) else nl) - 1         # rows l .. end belong to this launch
            if end == l:
                continue
            n_multi += 1
            ids = sorted({rows[r][key] for r in range(l, end + 1) for key in ("in_tensor", "out_tensor", "conv_tensor", "res_tensor")} - {-1})
            for i, a in enumerate(ids):
                assert tensors[a]["first_row"] <= l and tensors[a]["last_row"] >= min(end, nl), (l, end, a, tensors[a])
                for b in ids[i + 1:]:
                    assert not overlap(tensors[a], tensors[b]), (l, end, a, b)
    if name == "resnet50":
        assert n_multi >= (10 if batch >= 12 else 2)              # (small batches: few fused launches, by the library's own rules)


def test_feeder_reports_every_error_and_never_hangs(monkeypatch):
    """tf2_amd/feeder.py without a GPU: a thread whose set-up fails (no device here: torch.cuda.set_device raises) must not leave
    drain() waiting for ever, and errors of several threads are all reported."""
    from tf2_amd.feeder import StreamFeeder
    import contextlib, types, sys as _sys
    fake = types.SimpleNamespace(cuda=types.SimpleNamespace(set_device=lambda d: None, stream=lambda s: contextlib.nullcontext()))
    monkeypatch.setitem(_sys.modules, "torch", fake)
    f = StreamFeeder([object(), object()], ["r0", "r1"], "cpu")
    seen = []
    f.submit(0, lambda rn: seen.append(rn))
    f.submit(1, lambda rn: (_ for _ in ()).throw(ValueError("boom 1")))
    f.submit(0, lambda rn: (_ for _ in ()).throw(ValueError("boom 0")))
    with pytest.raises(RuntimeError, match="2 errors"):
        f.drain()
    assert seen == ["r0"]
    f.drain()                                    # errors are reported once
    f.close()
    # a thread that dies in its set-up releases drain()
    fake.cuda.set_device = lambda d: (_ for _ in ()).throw(RuntimeError("no device"))
    g = StreamFeeder([object()], ["r"], "cpu")
    g.submit(0, lambda rn: None)
    with pytest.raises(RuntimeError, match="no device"):
        g.drain()
    g.close()


def test_option_string_is_parsed_once_and_checked(monkeypatch):
    """csrc/opts.h: ONE option string (TF2_AMD_OPTS), unknown names and test-only options without TF2_AMD_TEST=1 are errors of
    tf2_net_create / tf2_net_reload_options (status codes, never exit), and no source file but opts.cpp reads the environment."""
    t = cfg.tiny_tables()
    monkeypatch.delenv("TF2_AMD_TEST", raising=False)
    monkeypatch.setenv("TF2_AMD_OPTS", "bband=0,c3=0")                    # product options: no TF2_AMD_TEST needed
    net = network.NetWork(t)
    monkeypatch.setenv("TF2_AMD_OPTS", "no_such_option=1")
    with pytest.raises(_lib.Tf2Error, match="unknown option"):
        net.reload_options()
    with pytest.raises(_lib.Tf2Error, match="unknown option"):
        network.NetWork(t)
    monkeypatch.setenv("TF2_AMD_OPTS", "nofast=1")
    with pytest.raises(_lib.Tf2Error, match="test-only"):
        net.reload_options()
    monkeypatch.setenv("TF2_AMD_TEST", "1")
    net.reload_options()
    monkeypatch.setenv("TF2_AMD_OPTS", "bband")                           # a bare name = 1
    net.reload_options()
    # round-5 advice: the separators the C parser documents (',', ';', ' ') and bare flags, through the Python helpers as well; 'name=' is an error
    monkeypatch.setenv("TF2_AMD_OPTS", "nodbl;fc=0 bband=2,share=1")
    net.reload_options()
    assert _lib.parse_opts("nodbl;fc=0 bband=2,share=1") == {"nodbl": "1", "fc": "0", "bband": "2", "share": "1"}
    _lib.set_opts(c3=0)
    assert _lib.parse_opts(os.environ["TF2_AMD_OPTS"])["c3"] == "0" and _lib.parse_opts(os.environ["TF2_AMD_OPTS"])["nodbl"] == "1"
    net.reload_options()
    monkeypatch.setenv("TF2_AMD_OPTS", "share=")
    with pytest.raises(_lib.Tf2Error, match="no value"):
        net.reload_options()
    # ... and the 56 variables of rounds 1-4 are not silently ignored any more: the error names the replacement
    monkeypatch.setenv("TF2_AMD_OPTS", "bband=1")
    monkeypatch.setenv("TF2_AMD_BGROUP", "0")
    with pytest.raises(_lib.Tf2Error, match=r'TF2_AMD_BGROUP is no longer read: use TF2_AMD_OPTS="bgroup=0"'):
        net.reload_options()
    with pytest.raises(_lib.Tf2Error, match="no longer read"):
        network.NetWork(t)
    monkeypatch.delenv("TF2_AMD_BGROUP")
    net.reload_options()
    csrc = os.path.join(ROOT, "tf2_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")) and f != "opts.cpp":
            assert not re.search(r"\bgetenv\s*\(\s*\"", open(os.path.join(csrc, f)).read()), f


def test_a_second_reader_of_the_image_keeps_the_plain_first_layer():
    """Net::init turns a 3x3 first layer on the 3-channel image into a pointwise layer over the im2col image -- the input TENSOR then
    holds im2col bytes.  A program in which another row reads the image as well must keep the plain form (round-4 advisor finding)."""
    t = cfg.vgg16_tables(32, 10)
    one = network.NetWork(t)
    q = synth.synth_q_values(t, 3)
    one.Init(synth.synth_model(t, q, 3), synth.q_text(q))
    assert "im2col" in one.describe_launches(2, 0)[0]["kernel"]
    t2 = cfg.vgg16_tables(32, 10)
    plan = cfg.build_plan(t2)
    if any(L.src == -1 for L in plan[1:]):
        pytest.skip("the builder already has a second image reader")
    # a second consumer of the image: row 1 rewired to read it (same shape as row 0's input: 3 channels, same map)
    descs = network._layer_descs(t2)
    plan2, nd, arr = descs
    import ctypes as C2
    arr[1].src, arr[1].C, arr[1].model_C, arr[1].q_in_row = -1, 3, 3, arr[0].q_in_row
    h = C2.c_void_p()
    _lib.check(_lib.lib().tf2_net_create(C2.byref(nd), arr, C2.byref(h)))
    try:
        q2 = np.zeros((nd.n_q_rows, nd.max_out_channel), np.int8)
        _lib.check(_lib.lib().tf2_net_set_q(h, q2.ctypes.data, q2.size))
        # an all-zero model stream of the right length: row 1 now has 3 input channels
        n_f = sum((L.N * (3 if i == 1 else L.model_C) * L.model_k * L.model_k) + (L.N if L.bias_en else 0) + (L.N * 4 + 1 if L.bn_en else 0) for i, L in enumerate(plan2) if not L.ipool)
        model = np.zeros(n_f, np.float32)
        _lib.check(_lib.lib().tf2_net_load_model(h, model.ctypes.data, model.size))
        _lib.check(_lib.lib().tf2_net_pack(h, 0))
        rows = (_lib.LaunchInfo * 512)()
        n = C2.c_int(0)
        _lib.check(_lib.lib().tf2_net_describe_launches(h, 2, 0, rows, 512, C2.byref(n)))
        assert "im2col" not in rows[0].kernel.decode(), rows[0].kernel
    finally:
        _lib.lib().tf2_net_destroy(h)
