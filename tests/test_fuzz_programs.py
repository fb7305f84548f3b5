"""Randomised layer programs through the C-ABI: every layer of seeded random networks -- first layers of both strides with and without
a pool, pointwise / 3x3 / strided / dilated rows, bottleneck triples with residuals, independent pools, odd map sizes and channel
counts -- against the oracle (GPU), and (without a device) that every such program packs and yields a launch plan in both modes.
The shapes are drawn so that the kernel SELECTION code is exercised on geometry the BASELINE networks never produce: maps that do not
divide into tiles, channel counts that are not multiples of 64, rows that qualify for a fused launch by a hair or miss it by one."""
import os
import sys

import numpy as np
import pytest

from tf2_amd import config as cfg, network, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SEEDS = list(range(56))


def random_program(seed: int) -> cfg.NetTables:
    rng = np.random.default_rng(9000 + seed)
    hw = int(rng.integers(17, 81))
    b = cfg._B(f"fuzz{seed}", image=(3, hw, hw), first_filter=3)
    widths = [16, 24, 32, 48, 64, 96, 128, 192, 256]

    def out_hw(h, k, s, p, d=1):
        return (h + 2 * p - d * (k - 1) - 1) // s + 1

    def maybe_pool(h):
        r = rng.random()
        if r < 0.25 and h >= 5:
            pad = int(rng.integers(0, 2))
            ph = -(-(h + 2 * pad - 3) // 2) + 1                     # ceil mode
            if (ph - 1) * 2 - pad >= h:
                ph -= 1
            return (3, 2, pad, ph, ph), ph
        if r < 0.4 and h >= 4:
            return (2, 2, 0, h // 2, h // 2), h // 2
        return None, h

    # first layer: 3x3 on the image, stride 1 or 2, pad 0 or 1, sometimes pooled
    s0, p0 = int(rng.integers(1, 3)), int(rng.integers(0, 2))
    if s0 == 2 and hw % 2 == 0:
        hw += 1          # (stride-2 windows then cover every column: tf2_net_read_layer(-1) rebuilds the image from the im2col tensor, which
                         #  holds no pixel that no window reads)
        b.image = (3, hw, hw)
    n0 = int(rng.choice([16, 32, 64]))
    h = out_hw(hw, 3, s0, p0)
    pool, hp = maybe_pool(h)
    cur = b.conv(-1, 3, hw, hw, n0, 3, s0, p0, relu=1, pool=pool)
    C, H = n0, hp
    n_blocks = int(rng.integers(3, 9))
    for _ in range(n_blocks):
        kind = rng.choice(["pw", "c3", "c3s2", "dil", "bottleneck", "bottleneck_proj", "ipool", "l2norm"])
        if kind == "pw":
            N = int(rng.choice(widths)); pool, hp = maybe_pool(H)
            cur = b.conv(cur, C, H, H, N, 1, 1, 0, relu=int(rng.integers(0, 2)) if pool is None else 1, pool=pool)
            C, H = N, hp
        elif kind == "c3":
            N = int(rng.choice(widths)); pool, hp = maybe_pool(H)
            cur = b.conv(cur, C, H, H, N, 3, 1, 1, relu=1, pool=pool, bias=int(rng.integers(0, 2)), bn=1)
            C, H = N, hp
        elif kind == "c3s2" and H >= 6:
            N = int(rng.choice(widths))
            cur = b.conv(cur, C, H, H, N, 3, 2, 1, relu=1)
            C, H = N, out_hw(H, 3, 2, 1)
        elif kind == "dil" and H >= 7:
            N = int(rng.choice(widths))
            cur = b.conv(cur, C, H, H, N, 3, 1, 2, relu=1, dil=2)
            C = N
        elif kind in ("bottleneck", "bottleneck_proj"):
            mid = int(rng.choice([16, 32, 64, 128])); s = 2 if (kind == "bottleneck_proj" and H >= 6 and rng.random() < 0.5) else 1
            out = C if kind == "bottleneck" else int(rng.choice(widths))
            Ho = out_hw(H, 3, s, 1)
            sc = cur if kind == "bottleneck" else b.conv(cur, C, H, H, out, 1, s, 0, relu=0)
            if kind == "bottleneck" and s != 1:
                continue
            a = b.conv(cur, C, H, H, mid, 1, 1, 0, relu=1)
            m = b.conv(a, mid, H, H, mid, 3, s, 1, relu=1)
            cur = b.conv(m, mid, Ho, Ho, out, 1, 1, 0, relu=0, add=sc, add_relu=1)
            C, H = out, Ho
        elif kind == "l2norm" and seed >= 40:              # (SSD's conv4_3 row; seeds below 40 keep the programs they always drew)
            cur = b.l2norm(cur, C, H, H)
        elif kind == "ipool" and H >= 5:
            ph = (H + 2 - 3) // 2 + 1
            cur = b.pool_only(cur, C, H, H, (3, 2, 1, ph, ph))
            H = ph
    # head: global average over the final map + a classifier
    cur = b.conv(cur, C, H, H, 64, 1, 1, 0, relu=0, endpool=1, endpool_hw=H * H) if H > 1 else cur
    b.conv(cur, 64 if H > 1 else C, 1, 1, 10, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


def _q_and_model(t, seed):
    q = synth.synth_q_values(t, seed, spread=int(seed % 3))
    return q, synth.synth_model(t, q, seed)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_programs_pack_and_plan(seed):
    """No device: the program parses, packs (mode 0) and both launch plans exist at three batch sizes; every conv row is covered by
    exactly one launch chain (no row skipped, none issued twice)."""
    t = random_program(seed)
    q, model = _q_and_model(t, seed)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    n_rows = len(cfg.build_plan(t))
    for batch in (1, 3, 32):
        for conc in (0, 1):
            rows = net.describe_launches(batch, conc)
            assert rows and all(r["grid"] >= 1 for r in rows)
            firsts = [r["layer"] for r in rows]
            assert firsts == sorted(firsts) and firsts[-1] <= n_rows - 1, firsts


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_programs_every_layer_against_the_oracle(seed):
    from test_gpu_parity import Rig
    t = random_program(seed)
    q, model = _q_and_model(t, seed)
    rig = Rig(t, q, model, 0)
    b = 1 + seed % 4
    x = synth.synth_images(t, b, seed, kind="int8" if seed % 2 else "float")
    want = rig.check_all_layers(x)
    # the plan of a plain run (nothing kept), one batch at a time and as if batches were in flight
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    for conc in (0, 1):
        got = rig.runner.run_batch(xd, concurrency=conc)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"plain run, concurrency {conc}")


# ---- the fused / specialised kernels on geometry of the test's choosing (their thresholds lowered through test-only options) -------------
def random_body_program(seed: int) -> cfg.NetTables:
    """VGG-like 3x3 rows (conv_c3 / conv_c3_w9, pooled and not), ResNet-like (3x3 + expand + residual) pairs (conv_bneck), a whole-window
    layer (conv_fc with 4-bit codes) -- at map sizes and channel counts drawn at random, 64-byte-aligned channels so that the kernels
    qualify."""
    rng = np.random.default_rng(7000 + seed)
    hw = int(rng.integers(20, 71))
    b = cfg._B(f"body{seed}", image=(3, hw, hw), first_filter=3)
    C = int(rng.choice([64, 128]))
    pool = None
    H = hw
    if rng.random() < 0.5:
        pool, H = (2, 2, 0, hw // 2, hw // 2), hw // 2
    cur = b.conv(-1, 3, hw, hw, C, 3, 1, 1, relu=1, pool=pool, bias=1, bn=0)
    for _ in range(int(rng.integers(2, 6))):
        kind = rng.choice(["c3", "c3pool", "bneck", "bneck"])
        if kind in ("c3", "c3pool"):
            N = int(rng.choice([64, 128, 256]))
            pool, hp = None, H
            if kind == "c3pool" and H >= 8:
                pool, hp = (2, 2, 0, H // 2, H // 2), H // 2
            cur = b.conv(cur, C, H, H, N, 3, 1, 1, relu=1, pool=pool, bias=int(rng.integers(0, 2)), bn=int(rng.integers(0, 2)))
            C, H = N, hp
        else:
            mid = int(rng.choice([64, 128]))
            out = 4 * mid
            sc = b.conv(cur, C, H, H, out, 1, 1, 0, relu=0) if C != out else cur
            a = b.conv(cur, C, H, H, mid, 1, 1, 0, relu=1)
            m = b.conv(a, mid, H, H, mid, 3, 1, 1, relu=1)
            cur = b.conv(m, mid, H, H, out, 1, 1, 0, relu=0, add=sc, add_relu=1)
            C = out
    # a whole-window layer (k = H, pad 0 -> 1 x 1) with a long K, then the classifier
    if H <= 9 and C * H * H >= 8 * 64:
        cur = b.conv(cur, C, H, H, 128, H, 1, 0, relu=1, bias=1, bn=0)
        C, H = 128, 1
    else:
        cur = b.conv(cur, C, H, H, 64, 1, 1, 0, relu=0, endpool=1, endpool_hw=H * H)
        C, H = 64, 1
    b.conv(cur, C, 1, 1, 16, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


BODY_SEEDS = list(range(24))
_BODY_OPTS = dict(c3_min="1", c3_min256="1", bneck_min="1", fc_min="8")


@pytest.mark.parametrize("seed", BODY_SEEDS)
def test_random_body_programs_pack_and_plan(seed, monkeypatch):
    from tests.conftest import set_opts
    set_opts(monkeypatch, **_BODY_OPTS)
    t = random_body_program(seed)
    q, model = _q_and_model(t, seed)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    for batch in (1, 2, 32):
        for conc in (0, 1):
            assert net.describe_launches(batch, conc)


def test_random_body_programs_reach_the_specialised_kernels(monkeypatch):
    """(the generator is only worth its name if the kernels it aims at are selected)"""
    from tests.conftest import set_opts
    set_opts(monkeypatch, **_BODY_OPTS)
    seen = set()
    for seed in BODY_SEEDS:
        t = random_body_program(seed)
        q, model = _q_and_model(t, seed)
        net = network.NetWork(t)
        net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
        seen |= {r["kernel"].split("<")[0].split(" ")[0] for r in net.describe_launches(2, 1)}
    assert {"conv_c3_kernel", "conv_bneck_kernel", "conv_first_kernel"} <= seen, seen
    assert any(k.startswith("fc") for k in seen), seen


@pytest.mark.gpu
@pytest.mark.parametrize("seed", BODY_SEEDS)
def test_random_body_programs_every_layer_against_the_oracle(seed, monkeypatch):
    from test_gpu_parity import Rig
    from tests.conftest import set_opts
    set_opts(monkeypatch, **_BODY_OPTS)
    t = random_body_program(seed)
    q, model = _q_and_model(t, seed)
    rig = Rig(t, q, model, 0)
    b = 1 + seed % 3
    x = synth.synth_images(t, b, seed, kind="int8" if seed % 2 else "float")
    want = rig.check_all_layers(x)
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    for conc in (0, 1):
        got = rig.runner.run_batch(xd, concurrency=conc)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"plain run, concurrency {conc}")


# ---- SqueezeNet-shaped programs: fire modules of random widths on the 56 / 28 / 14 maps conv_fire.hip takes (fire=1: wherever it fits) ----
def random_fire_program(seed: int) -> cfg.NetTables:
    rng = np.random.default_rng(5000 + seed)
    n1 = int(rng.choice([64, 128]))
    fires = []
    h = 56
    for _ in range(int(rng.integers(2, 6))):
        sq, ex = int(rng.choice([16, 32, 48, 64])), int(rng.choice([64, 128, 192, 256]))
        pool_after = bool(h > 14 and rng.random() < 0.4)
        fires.append((sq, ex, pool_after))
        if pool_after:
            h //= 2
    return cfg.fire_net_tables(56, (n1, 3, 1, 1, False), tuple(fires), 40, 16, f"fire{seed}")


FIRE_SEEDS = list(range(12))


def test_random_fire_programs_reach_the_fire_kernel(monkeypatch):
    from tests.conftest import set_opts
    set_opts(monkeypatch, fire="1")
    shapes = set()
    for seed in FIRE_SEEDS:
        t = random_fire_program(seed)
        q = synth.synth_q_values(t, 6, spread=1)                 # (bench.py's draw: one-window merged expands)
        net = network.NetWork(t)
        net.Quantization(synth.q_text(q)); net.LoadModel(synth.synth_model(t, q, 6)); net.Pack(0)
        for conc in (0, 1):
            rows = net.describe_launches(3, conc)
            assert rows
            shapes |= {r["kernel"].split(" (")[0] for r in rows if "conv_fire" in r["kernel"]}
    assert len(shapes) >= 6, shapes                              # several (map, C, S, N) instantiations / geometries


@pytest.mark.gpu
@pytest.mark.parametrize("seed", FIRE_SEEDS)
def test_random_fire_programs_every_layer_against_the_oracle(seed, monkeypatch):
    from test_gpu_parity import Rig
    from tests.conftest import set_opts
    set_opts(monkeypatch, fire="1")
    t = random_fire_program(seed)
    q = synth.synth_q_values(t, 6, spread=1)
    rig = Rig(t, q, synth.synth_model(t, q, 6), 0)
    b = 1 + seed % 3
    x = synth.synth_images(t, b, seed, kind="int8" if seed % 2 else "float")
    want = rig.check_all_layers(x)
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    for conc in (0, 1):
        got = rig.runner.run_batch(xd, concurrency=conc)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"plain run, concurrency {conc}")


# ---- the same random programs through the other packed forms and forced kernels --------------------------------------------------------
_VARIANTS = [("mode", 1), ("mode", 2), ("nofast", "1"), ("nodual", "1"), ("nodbl", "1"), ("nosemi", "1"), ("sk", "1"), ("sk", "2"), ("pw", "0"),
             ("im2col0", "0"), ("first", "0"), ("first_pool", "0"), ("dense", "0"), ("nofuse", "1"), ("share", "0"), ("no4bit", "1")]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(32))
def test_random_programs_other_packed_forms(i, monkeypatch):
    """pack modes 1 (the north-star split: k > 1 on the shift kernel) and 2 (shift kernel everywhere), generic requantisation, Horner
    windows, no doubled channels, forced / forbidden split-K, no pointwise kernel, plain first layers, table gathers, ...: one switch per
    case, two random programs each."""
    from test_gpu_parity import Rig
    from tests.conftest import set_opts
    name, val = _VARIANTS[i % len(_VARIANTS)]
    seed = 100 + i
    mode = 0
    if name == "mode":
        mode = val
    else:
        set_opts(monkeypatch, **{name: val})
    t = random_program(seed) if i % 3 else random_body_program(seed)
    q, model = _q_and_model(t, seed)
    rig = Rig(t, q, model, mode)
    x = synth.synth_images(t, 1 + i % 3, seed, kind="int8" if i % 2 else "float")
    rig.check_all_layers(x)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 3, 5, 8, 13, 21])
@pytest.mark.parametrize("graph", [False, True])
def test_random_body_programs_with_four_batches_in_flight(seed, graph, monkeypatch):
    """The LDS-DMA kernels at drawn geometry under the load they are built for: four runners of one handle on four streams, eight rotating
    inputs, 24 steps with no synchronisation in between (launched / replayed from HIP graphs), every step of the last four against a serial
    run, the serial run against the oracle."""
    from test_gpu_configs import _in_flight
    from test_gpu_parity import Rig
    from tests.conftest import set_opts
    set_opts(monkeypatch, **_BODY_OPTS)
    t = random_body_program(seed)
    q, model = _q_and_model(t, seed)
    rig = Rig(t, q, model, 0)
    _in_flight(rig, 3 + seed % 3, 8, 24, graph, 400 + seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,batch", [(2, 32), (9, 17), (16, 33), (20, 64)])
def test_random_body_programs_at_full_batches(seed, batch, monkeypatch):
    """... at batches that fill the chip (grids in several rounds, persistent blocks walking several tiles, conv_fc's chunks of 32 images):
    logits of every image against the oracle, both launch plans."""
    from test_gpu_parity import Rig
    from tests.conftest import set_opts
    set_opts(monkeypatch, **_BODY_OPTS)
    t = random_body_program(seed)
    q, model = _q_and_model(t, seed)
    rig = Rig(t, q, model, 0)
    x = synth.synth_images(t, batch, seed)
    want = rig.ref.logits(rig.ref.run(x))
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    for conc in (0, 1):
        got = rig.runner.run_batch(xd, concurrency=conc)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"concurrency {conc}")


# ---- inception-shaped programs: concat tensors written by branch tails, independent stride-1 pools, 5x5 and dilated branches --------------
def random_inception_program(seed: int) -> cfg.NetTables:
    rng = np.random.default_rng(3000 + seed)
    hw = int(rng.integers(20, 57))
    b = cfg._B(f"incep{seed}", image=(3, hw, hw), first_filter=3)
    C = int(rng.choice([32, 64]))
    H = hw
    pool = None
    if rng.random() < 0.5:
        ph = -(-(hw - 3) // 2) + 1
        pool, H = (3, 2, 0, ph, ph), ph
    cur = b.conv(-1, 3, hw, hw, C, 3, 1, 1, relu=1, pool=pool)
    widths = [16, 32, 48, 64, 96]
    cid = 0
    for _ in range(int(rng.integers(2, 5))):
        n0 = 0
        n1 = int(rng.choice(widths))
        b.conv(cur, C, H, H, n1, 1, 1, 0, relu=1, cat=(cid, n0, n0 + n1)); n0 += n1
        r2, n2 = int(rng.choice([16, 24, 32, 64])), int(rng.choice(widths))
        a = b.conv(cur, C, H, H, r2, 1, 1, 0, relu=1)
        b.conv(a, r2, H, H, n2, 3, 1, 1, relu=1, cat=(cid, n0, n0 + n2)); n0 += n2
        if rng.random() < 0.6 and H >= 7:
            r3, n3 = int(rng.choice([16, 24, 32])), int(rng.choice([16, 32, 48]))
            a = b.conv(cur, C, H, H, r3, 1, 1, 0, relu=1)
            if rng.random() < 0.5:
                b.conv(a, r3, H, H, n3, 5, 1, 2, relu=1, cat=(cid, n0, n0 + n3))
            else:
                b.conv(a, r3, H, H, n3, 3, 1, 2, relu=1, dil=2, cat=(cid, n0, n0 + n3))
            n0 += n3
        if rng.random() < 0.7:
            n4 = int(rng.choice([16, 32, 64]))
            p = b.pool_only(cur, C, H, H, (3, 1, 1, H, H))
            b.conv(p, C, H, H, n4, 1, 1, 0, relu=1, cat=(cid, n0, n0 + n4)); n0 += n4
        cur, C = ("C", cid), n0
        cid += 1
        if rng.random() < 0.4 and H >= 8:
            ph = -(-(H - 3) // 2) + 1
            cur = b.pool_only(cur, C, H, H, (3, 2, 0, ph, ph))
            H = ph
    cur = b.conv(cur, C, H, H, 32, 1, 1, 0, relu=0, endpool=1, endpool_hw=H * H)
    b.conv(cur, 32, 1, 1, 12, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


INCEPTION_SEEDS = list(range(20))


@pytest.mark.parametrize("seed", INCEPTION_SEEDS)
def test_random_inception_programs_pack_and_plan(seed):
    t = random_inception_program(seed)
    q, model = _q_and_model(t, seed)
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
    for batch in (1, 4, 32):
        for conc in (0, 1):
            assert net.describe_launches(batch, conc)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", INCEPTION_SEEDS)
def test_random_inception_programs_every_layer_against_the_oracle(seed):
    from test_gpu_parity import Rig
    t = random_inception_program(seed)
    q, model = _q_and_model(t, seed)
    rig = Rig(t, q, model, 0)
    x = synth.synth_images(t, 1 + seed % 3, seed, kind="int8" if seed % 2 else "float")
    want = rig.check_all_layers(x)
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    for conc in (0, 1):
        got = rig.runner.run_batch(xd, concurrency=conc)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"plain run, concurrency {conc}")


def test_every_random_program_plans_at_every_batch_size():
    """No device: both launch plans of all the generators' programs at batch 1..19 and around the powers of two up to 3000 (beyond the fused
    first layers' 32-bit pixel indices): a plan exists, covers the rows in order and names a kernel for every launch."""
    progs = ([(s, random_program(s)) for s in SEEDS[::4]] + [(s, random_body_program(s)) for s in BODY_SEEDS[::3]] +
             [(s, random_fire_program(s)) for s in FIRE_SEEDS[::3]] + [(s, random_inception_program(s)) for s in INCEPTION_SEEDS[::4]])
    for seed, t in progs:
        q, model = _q_and_model(t, seed)
        net = network.NetWork(t)
        net.Quantization(synth.q_text(q)); net.LoadModel(model); net.Pack(0)
        for batch in list(range(1, 20)) + [31, 32, 33, 63, 64, 65, 128, 255, 256, 700, 3000]:
            for conc in (0, 1):
                rows = net.describe_launches(batch, conc)
                assert rows and all(r["kernel"] and r["grid"] >= 1 for r in rows), (seed, batch, conc)
                layers = [r["layer"] for r in rows]
                assert layers == sorted(layers), (seed, batch, conc, layers)


@pytest.mark.parametrize("which,seed", [("free", s) for s in range(0, 56, 3)] + [("body", s) for s in range(0, 24, 3)] +
                         [("inception", s) for s in range(0, 20, 3)] + [("fire", s) for s in range(0, 12, 3)])
@pytest.mark.parametrize("mode", [0, 2])
def test_random_programs_packed_image_emulated_on_the_cpu(which, seed, mode, monkeypatch):
    """No device: the packed weight image of a random program (exponent windows, doubled channels, slab lists, 4-bit code layers, merged rows,
    the im2col first layer) run through the numpy model of the kernels' data flow (tests/emu_packed.py) reproduces the oracle layer by layer --
    mode 0 (MFMA forms) and mode 2 (the shift kernel's packed 4-bit filters)."""
    from tests.conftest import set_opts
    from tests.test_pack_emulation import check_net
    if which == "body":
        set_opts(monkeypatch, **_BODY_OPTS)
    t = {"free": random_program, "body": random_body_program, "inception": random_inception_program, "fire": random_fire_program}[which](seed)
    q, model = _q_and_model(t, seed)
    if which == "fire":
        q = synth.synth_q_values(t, 6, spread=1)                 # (one-window merged expands: the merged rows are in the packed image)
        model = synth.synth_model(t, q, 6)
    x = synth.synth_images(t, 2, seed, kind="int8" if seed % 2 else "float")
    if seed % 2:
        x[0, :, :2, :] = -128                                    # the negate quirk (pe.cl:32-37)
    check_net(t, q, model, x, mode)
