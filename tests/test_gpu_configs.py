"""GPU parity at the BASELINE.json configurations and on the run paths round 1 left unexercised: ResNet50 batch 64,
the reference's third shipped network (pruned ResNet50: channel counts that are not multiples of 64), SSD300 at full
width, VGG16 at its per-GPU batch, the HIP-graph replay path and the 4-bit packed model file on the GPU."""
import json
import os

import numpy as np
import pytest

from tf2_amd import config as cfg, model4bit, network, synth

import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_parity import Rig, _torch  # noqa: E402

from tests.conftest import set_opts  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def r50(golden_dir):
    t = cfg.resnet50_tables()
    q = np.loadtxt(os.path.join(golden_dir, "resnet50_Q"), dtype=np.int32)
    return t, q, synth.synth_model(t, q, 0)


@pytest.fixture(scope="module")
def r50_rig(r50):
    return Rig(*r50, 0)


def test_resnet50_batch64_logits_vs_oracle(r50_rig):
    """BASELINE configs[2]: ResNet50 batch 64 on one GPU -- all 64 logits rows and top-5 lists against the oracle."""
    x = synth.synth_images(r50_rig.t, 64, 64)
    got = r50_rig.run(x, keep_all=False)
    want = r50_rig.ref.logits(r50_rig.ref.run(x))
    np.testing.assert_array_equal(got, want)
    for b in (0, 31, 63):
        lab, _ = network.Evaluation(b, r50_rig.net.q, got, num_layer=r50_rig.net.num_layer)
        assert lab == r50_rig.ref.top5(want[b])[0].tolist()


def test_resnet50_pruned_every_layer(golden_dir):
    """host/inc/resnet50_pruned.h + host/model/resnet50_pruned_Q (cnn.h:29-35): the shipped tables of the channel-pruned
    network, widths 32..448 that are not multiples of the 64-row MFMA tiles -- every layer against the oracle."""
    t = cfg.NetTables(json.load(open(os.path.join(golden_dir, "tables_resnet50_pruned.json"))))
    t.setdefault("xConv1Rewrite", 1)
    qv = np.loadtxt(os.path.join(golden_dir, "resnet50_pruned_Q"), dtype=np.int32)
    model = synth.synth_model(t, qv, 13)
    rig = Rig(t, qv, model, 0)
    widths = {L.N for L in rig.ref.plan}
    assert any(w % 64 for w in widths)
    rig.check_all_layers(synth.synth_images(t, 3, 13))
    # and the logits at batch 32 (tile tails on every map size)
    x = synth.synth_images(t, 32, 14)
    np.testing.assert_array_equal(rig.run(x, keep_all=False), rig.ref.logits(rig.ref.run(x)))


def test_ssd300_full_width_and_batch32_properties():
    """BASELINE configs[4]: SSD300-VGG at full width.  Two images against the oracle on every row, then batch 32:
    rows 0-1 equal the oracle-checked pair, a batch permutation permutes the outputs, an image run in a pair equals
    its row in the batch."""
    import torch
    t = cfg.ssd300_tables()
    q = synth.synth_q_values(t, 3, spread=1)
    model = synth.synth_model(t, q, 3)
    rig = Rig(t, q, model, 0)
    x = synth.synth_images(t, 32, 5)
    rig.check_all_layers(x[:2])
    heads = [l for l, L in enumerate(rig.ref.plan) if l >= 24]
    rig.run(x, keep_all=True)
    full = {l: rig.runner.read_layer(l, 32) for l in heads}
    perm = np.random.default_rng(1).permutation(32)
    rig.run(x[perm], keep_all=True)
    for l in heads:
        np.testing.assert_array_equal(rig.runner.read_layer(l, 32), full[l][perm], err_msg=f"head row {l} under a batch permutation")
    rig.run(x[:2], keep_all=True)
    for l in heads:
        np.testing.assert_array_equal(rig.runner.read_layer(l, 2), full[l][:2])
    torch.cuda.synchronize()


def test_vgg16_batch32_properties():
    """BASELINE configs[3]: VGG16 at its per-GPU batch (256 over 8 GPUs = 32): two images against the oracle, the rest
    through batch invariance and permutation."""
    t = cfg.vgg16_tables()
    q = synth.synth_q_values(t, 1, spread=1)
    model = synth.synth_model(t, q, 1)
    rig = Rig(t, q, model, 0)
    x = synth.synth_images(t, 32, 6)
    got = rig.run(x, keep_all=False)
    np.testing.assert_array_equal(got[:2], rig.ref.logits(rig.ref.run(x[:2])))
    perm = np.random.default_rng(2).permutation(32)
    np.testing.assert_array_equal(rig.run(x[perm], keep_all=False), got[perm])
    np.testing.assert_array_equal(rig.run(x[20:22], keep_all=False), got[20:22])


@pytest.mark.parametrize("batch,conc", [(1, 0), (4, 0), (16, 0), (16, 1), (32, 0)])
def test_hip_graph_replay_equals_run_batch(r50_rig, batch, conc):
    """Runner.capture: the step recorded once into a HIP graph; replays on refilled input buffers give the logits of
    run_batch (and of the oracle).  Batches 16 and 32 captured as "one batch at a time" take the group launches (conv_bgroup.hip;
    batch 32: stage 4's five bottlenecks in one launch): their flags carry a step counter that the replayed input-preparation
    kernel advances, so a replay is as good as a launch.  conc = 1: the several-batches-in-flight plan, as bench.py replays it."""
    torch = _torch()
    rig = r50_rig
    if batch >= 12:
        assert any("conv_bgroup" in r["kernel"] for r in rig.net.describe_launches(batch, 0))
        assert not any("conv_bgroup" in r["kernel"] for r in rig.net.describe_launches(batch, 1))
    xs = [synth.synth_images(rig.t, batch, 70 + i) for i in range(3)]
    want = [rig.run(x, keep_all=False) for x in xs]
    np.testing.assert_array_equal(want[0][:2], rig.ref.logits(rig.ref.run(xs[0][:2])))
    runner = network.Runner(None, rig.net)
    buf = torch.from_numpy(xs[0]).to("cuda:0")
    replay = runner.capture(buf, concurrency=conc)
    for x, w in zip(xs, want):
        buf.copy_(torch.from_numpy(x))
        replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(runner._logits.cpu().numpy(), w)


def test_model4bit_file_to_gpu_logits(r50):
    """TransForm_Kit's 4-bit packed model file (4bit_data_format.txt) -> tf2_net_load_model_4bit -> GPU: the logits of
    the float32-loaded network, bit for bit, and the oracle's."""
    torch = _torch()
    t, q, model = r50
    blob = model4bit.write_model_4bit(t, model)
    assert len(blob) < model.nbytes / 5
    net = network.NetWork(t)
    net.Quantization(synth.q_text(q)); net.LoadModel4bit(blob); net.Pack(0); net.InitBuffer("cuda:0")
    x = synth.synth_images(t, 3, 9)
    got = network.Runner(None, net).run_batch(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
    rig = Rig(t, q, model, 0)
    np.testing.assert_array_equal(got, rig.run(x, keep_all=False))
    np.testing.assert_array_equal(got, rig.ref.logits(rig.ref.run(x)))


def test_run_split_and_its_graph_equal_run_batch(r50_rig):
    """Runner.run_split: the batch as 2 / 3 sub-batches on concurrent streams (fork / join on the caller's stream)
    writes the same logits tensor as run_batch; also when captured into a HIP graph."""
    torch = _torch()
    rig = r50_rig
    x = synth.synth_images(rig.t, 7, 91)
    want = rig.run(x, keep_all=False)
    xd = torch.from_numpy(x).to("cuda:0")
    for parts in (2, 3):
        r = network.Runner(None, rig.net)
        got = r.run_split(xd, parts)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    r = network.Runner(None, rig.net)
    buf = xd.clone()
    replay = r.capture(buf, split=2)
    x2 = synth.synth_images(rig.t, 7, 92)
    buf.copy_(torch.from_numpy(x2))
    replay(); torch.cuda.synchronize()
    np.testing.assert_array_equal(r._logits.cpu().numpy(), rig.run(x2, keep_all=False))


@pytest.mark.parametrize("conc", ["0", "1"])
def test_resnet50_batch_sizes_and_tile_choices(r50, monkeypatch, conc):
    """The launch plan picks tile heights by batch size and by whether batches are in flight (wide / narrow alternatives,
    fused or separate bottleneck pairs, conv_pw or the ring kernel): batch sizes on both sides of every threshold, with the
    one-stream and the several-streams choice forced, against the oracle (three images each) and against themselves (an image
    run alone gives its row of the batch)."""
    set_opts(monkeypatch, alt_conc=conc)
    rig = Rig(*r50, 0)
    x = synth.synth_images(rig.t, 64, 91)
    want = rig.ref.logits(rig.ref.run(x[:3]))
    rows = {}
    for b in (1, 2, 3, 4, 9, 17, 33, 64):
        got = rig.run(x[:b], keep_all=False)
        np.testing.assert_array_equal(got[:min(b, 3)], want[:min(b, 3)], err_msg=f"batch {b}")
        for i in range(b):
            if i in rows:
                np.testing.assert_array_equal(got[i], rows[i], err_msg=f"batch {b}, image {i}")
            else:
                rows[i] = got[i].copy()


@pytest.mark.parametrize("graph", [0, 1])
def test_four_batches_in_flight_as_bench_times_them(r50, graph):
    """The configuration bench.py's headline is timed in, checked as executed: ONE Net, four Runners (own workspaces) on four
    HIP streams (GPU_MAX_HW_QUEUES=8 from conftest), the library's stream-history heuristic left on auto, eight different
    batch-32 inputs issued round-robin for 16 steps with no synchronisation in between -- launched, and replayed from captured
    HIP graphs.  Every logits row of the last four steps: three rows per batch against the oracle, all 32 against a serial
    run_batch of the same input (feature_writer.cl:88-151 is the output contract of each step)."""
    torch = _torch()
    assert os.environ.get("GPU_MAX_HW_QUEUES") == "8"
    rig = Rig(*r50, 0)
    n_fl, n_in, n_steps = 4, 8, 32
    xs = [synth.synth_images(rig.t, 32, 300 + i) for i in range(n_in)]
    xd = [torch.from_numpy(x).to("cuda:0") for x in xs]
    serial = [rig.run(x, keep_all=False).copy() for x in xs]          # one stream, one batch at a time
    for i in range(n_in - n_fl, n_in):
        np.testing.assert_array_equal(serial[i][:3], rig.ref.logits(rig.ref.run(xs[i][:3])), err_msg=f"serial run, input {i}")
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(n_fl)]
    runners = [network.Runner(None, rig.net) for _ in range(n_fl)]
    bufs = [xd[i].clone() for i in range(n_fl)]                        # static input buffers of the captured graphs
    for st, rn, b in zip(streams, runners, bufs):                      # bench.py timed(): set-up call per stream
        with torch.cuda.stream(st):
            rn.run_batch(b)
    torch.cuda.synchronize()
    replays = [None] * n_fl
    for k in range(n_steps):                                           # no synchronisation inside this loop
        i = k % n_fl
        with torch.cuda.stream(streams[i]):
            if graph:
                if replays[i] is None:
                    replays[i] = runners[i].capture(bufs[i])
                bufs[i].copy_(xd[k % n_in], non_blocking=True)
                replays[i]()
            else:
                runners[i].run_batch(xd[k % n_in])
    torch.cuda.synchronize()
    for k in range(n_steps - n_fl, n_steps):
        got = runners[k % n_fl]._logits.cpu().numpy()
        np.testing.assert_array_equal(got, serial[k % n_in], err_msg=f"step {k} on stream {k % n_fl} (graph={graph})")


def _in_flight(rig, B, n_in, n_steps, graph, seed, oracle_images=1, keep_rows=None, n_fl=4):
    """`n_fl` Runners of ONE Net on `n_fl` HIP streams, `n_in` different batch-`B` inputs issued round-robin for `n_steps` steps with
    no synchronisation in between (launched, or replayed from one captured HIP graph per stream): every logits row of the last `n_fl`
    steps against a serial run of the same input (feature_writer.cl:88-151 is the output contract of each step), the serial run of
    input 0 against the oracle.  keep_rows: table rows read back as well (a keep_all workspace: SSD300's heads are its outputs)."""
    torch = _torch()
    t = rig.t
    xs = [synth.synth_images(t, B, seed + i) for i in range(n_in)]
    xd = [torch.from_numpy(x).to("cuda:0") for x in xs]
    keep = keep_rows is not None
    serial, serial_rows = [], []
    for x in xs:
        serial.append(rig.run(x, keep_all=keep).copy())
        serial_rows.append({l: rig.runner.read_layer(l, B) for l in (keep_rows or [])})
    outs = rig.ref.run(xs[0][:oracle_images])
    np.testing.assert_array_equal(serial[0][:oracle_images], rig.ref.logits(outs), err_msg="serial run against the oracle")
    for l in (keep_rows or []):
        np.testing.assert_array_equal(serial_rows[0][l][:oracle_images], outs[l], err_msg=f"serial run, row {l} against the oracle")
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(n_fl)]
    runners = [network.Runner(None, rig.net) for _ in range(n_fl)]
    bufs = [xd[i % n_in].clone() for i in range(n_fl)]
    for st, rn, b in zip(streams, runners, bufs):                      # set-up call per stream (workspace allocation)
        with torch.cuda.stream(st):
            rn.run_batch(b, keep_all=keep, concurrency=1)
    torch.cuda.synchronize()
    replays = [None] * n_fl
    for k in range(n_steps):                                           # no synchronisation inside this loop
        i = k % n_fl
        with torch.cuda.stream(streams[i]):
            if graph:
                if replays[i] is None:
                    replays[i] = runners[i].capture(bufs[i], concurrency=1)
                bufs[i].copy_(xd[k % n_in], non_blocking=True)
                replays[i]()
            else:
                runners[i].run_batch(xd[k % n_in], keep_all=keep, concurrency=1)
    torch.cuda.synchronize()
    for k in range(n_steps - n_fl, n_steps):
        i = k % n_fl
        np.testing.assert_array_equal(runners[i]._logits.cpu().numpy(), serial[k % n_in], err_msg=f"step {k} on stream {i} (graph={graph})")
        for l in (keep_rows or []):
            np.testing.assert_array_equal(runners[i].read_layer(l, B), serial_rows[k % n_in][l], err_msg=f"step {k} on stream {i}, row {l}")


def test_vgg16_four_batches_in_flight():
    """The VGG-style kernels under load: conv_c3 / conv_c3_w9 (counted waits on LDS-DMAs that land later when other batches' kernels
    share the chip), conv_fc (scratch area per workspace) and the im2col input kernel -- full-size VGG16, four Runners on four streams,
    batch 24, six inputs round-robin for 24 steps with no synchronisation; every logits row of the last four steps against a serial
    run of the same input, and one image per input against the oracle."""
    t = cfg.vgg16_tables()
    q = synth.synth_q_values(t, 0, spread=1)
    rig = Rig(t, q, synth.synth_model(t, q, 0), 0)
    names = [r["kernel"] for r in rig.net.describe_launches(24, 1)]
    assert any("conv_c3_w9" in n for n in names) and any("conv_c3_kernel" in n for n in names) and any("fc4_partial" in n or "fc_partial" in n for n in names), names
    _in_flight(rig, B=24, n_in=6, n_steps=24, graph=0, seed=500)


@pytest.mark.parametrize("graph", [0, 1])
def test_ssd300_four_batches_in_flight(graph):
    """BASELINE configs[4] under the load the product is built for: SSD300-VGG at full width (300 x 300 tiles of conv_c3 / conv_c3_w9,
    the stride-2 and dilated rows on the ring kernel, the L2Norm row, twelve heads), batch 8 on four streams.  Launched: a keep_all
    workspace per stream, every head row (the network's outputs) of the last four steps against serial runs; replayed from HIP
    graphs: the last head."""
    t = cfg.ssd300_tables()
    q = synth.synth_q_values(t, 3, spread=1)
    rig = Rig(t, q, synth.synth_model(t, q, 3), 0)
    names = [r["kernel"] for r in rig.net.describe_launches(8, 1)]
    assert any("conv_c3_w9" in n for n in names) and any("conv_c3_kernel" in n for n in names), names
    heads = [l for l, L in enumerate(rig.ref.plan) if l >= 24]
    _in_flight(rig, B=8, n_in=4, n_steps=16, graph=graph, seed=520, keep_rows=None if graph else heads)


@pytest.mark.parametrize("graph", [0, 1])
def test_squeezenet_four_batches_in_flight(graph):
    """BASELINE configs[1]: SqueezeNet 1.1 at 227 x 227 (im2col first layer on conv_pw, fire modules as concat slices, ceil-mode pools,
    the global average, the signed 1000 -> 128 classifier on conv_shift_fc), batch 32 on four streams, 32 steps."""
    t = cfg.squeezenet11_tables()
    q = synth.synth_q_values(t, 21, spread=2)
    rig = Rig(t, q, synth.synth_model(t, q, 21), 0)
    assert any("conv_shift_fc" in r["kernel"] for r in rig.net.describe_launches(32, 1))
    _in_flight(rig, B=32, n_in=8, n_steps=32, graph=graph, seed=540, oracle_images=2)


@pytest.mark.parametrize("graph", [0, 1])
def test_googlenet_four_batches_in_flight(graph, golden_dir):
    """The reference's second shipped network (googlenet.h tables, shipped googlenet_Q: ipool rows, four-way concat slices, 5x5 convs,
    pools distributed onto branch tails), batch 16 on four streams, 32 steps."""
    t = cfg.NetTables(json.load(open(os.path.join(golden_dir, "tables_googlenet.json"))))
    t.setdefault("xConv1Rewrite", 1)
    qv = np.loadtxt(os.path.join(golden_dir, "googlenet_Q"), dtype=np.int32)
    rig = Rig(t, qv, synth.synth_model(t, qv, 7), 0)
    _in_flight(rig, B=16, n_in=8, n_steps=32, graph=graph, seed=560, oracle_images=2)


@pytest.mark.parametrize("graph", [0, 1])
def test_resnet50_pruned_four_batches_in_flight(graph, golden_dir):
    """The reference's third shipped network (resnet50_pruned.h: widths 32..448, tile tails on every map), batch 32 on four streams."""
    t = cfg.NetTables(json.load(open(os.path.join(golden_dir, "tables_resnet50_pruned.json"))))
    t.setdefault("xConv1Rewrite", 1)
    qv = np.loadtxt(os.path.join(golden_dir, "resnet50_pruned_Q"), dtype=np.int32)
    rig = Rig(t, qv, synth.synth_model(t, qv, 13), 0)
    _in_flight(rig, B=32, n_in=8, n_steps=32, graph=graph, seed=580, oracle_images=2)


def test_feeder_threads_enqueue_side_by_side(r50):
    """tf2_amd/feeder.py: four host threads, one per stream and workspace, call tf2_net_run_ex on ONE handle at the same time
    (include/tf2_amd.h threading note: the enqueue runs outside the handle's mutex).  40 steps over eight inputs with no
    synchronisation; the last four steps' logits against serial runs of the same inputs; plain streams, as bench.py's."""
    torch = _torch()
    from tf2_amd.feeder import StreamFeeder
    rig = Rig(*r50, 0)
    n_fl, n_in, n_steps = 4, 8, 40
    xs = [synth.synth_images(rig.t, 32, 400 + i) for i in range(n_in)]
    xd = [torch.from_numpy(x).to("cuda:0") for x in xs]
    serial = [rig.run(x, keep_all=False).copy() for x in xs]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(n_fl)]
    runners = [network.Runner(None, rig.net) for _ in range(n_fl)]
    feeder = StreamFeeder(streams, runners, torch.device("cuda:0"))
    try:
        for k in range(n_steps):
            feeder.submit(k % n_fl, lambda rn, x=xd[k % n_in]: rn.run_batch(x, concurrency=1))
        feeder.drain()
        torch.cuda.synchronize()
        for k in range(n_steps - n_fl, n_steps):
            np.testing.assert_array_equal(runners[k % n_fl]._logits.cpu().numpy(), serial[k % n_in], err_msg=f"step {k} on feeder {k % n_fl}")
        feeder.submit(0, lambda rn: rn.run_batch(xd[0][:, :, :5]))            # an error on a feeder thread surfaces in drain()
        with pytest.raises(Exception):
            feeder.drain()
    finally:
        feeder.close()


def test_cli_on_the_shipped_image(r50, golden_dir, tmp_path, capsys):
    """The reference's host CLI (main.cpp:19-61: model_file q_file image_file verify_file num_images) through tf2_amd.cli: the
    shipped test image and Q file, a model file in the float32 LoadModel stream format, the shipped golden logits as the verify
    file.  Prints what main() prints (arguments, latency / throughput, the compare line, top-5 per image) and its top-5 equals
    the oracle's for the same model."""
    from tf2_amd import cli
    t, q, model = r50
    mf = tmp_path / "param.bin"
    np.asarray(model, np.float32).tofile(mf)
    img = os.path.join(golden_dir, "resnet50_data_label_100.bin")
    rc = cli.main([str(mf), os.path.join(golden_dir, "resnet50_Q"), img, os.path.join(golden_dir, "resnet50_fc1000_label_100.bin"), "2"])
    assert rc == 0
    out = capsys.readouterr().out
    assert "num_images = 2" in out and "Latency = " in out and "Throughput = " in out
    assert out.count("compare finished, error=") == 2 and out.count("rank=0") == 2
    labels = [int(l.split("label=")[1].split()[0]) for l in out.splitlines() if l.startswith("rank=")]
    assert len(labels) == 10 and labels[:5] == labels[5:]
    rig = Rig(t, q, model, 0)
    x = np.fromfile(img, np.float32).reshape(1, 3, 224, 224)
    want = rig.ref.logits(rig.ref.run(x))
    assert labels[:5] == rig.ref.top5(want[0])[0].tolist()
    # a missing verify file is reported, not fatal (the reference prints the open error and goes on, network_helper.cpp:77-83)
    assert cli.main([str(mf), os.path.join(golden_dir, "resnet50_Q"), img, str(tmp_path / "nope.bin"), "1"]) == 0
    assert "verify file not readable" in capsys.readouterr().out


def test_group_launch_preconditions_are_checked_per_call(r50):
    """include/tf2_amd.h, group launches: (1) a stream whose CU mask leaves fewer than 64 CUs never takes them, even when the caller
    states concurrency = 0 -- the step runs with separate launches and the same logits; (2) five host threads, one stream and
    workspace each, calling tf2_net_run on ONE handle: the library sees the several streams in its call history and selects the
    in-flight plan -- no group launch is ever enqueued there (tf2_net_run_stats), and every thread gets the logits of a serial run."""
    torch = _torch()
    import threading
    from tf2_amd import streams as tstreams
    rig = Rig(*r50, 0)
    x = torch.from_numpy(synth.synth_images(rig.t, 32, 411)).to("cuda:0")
    want = rig.runner.run_batch(x, concurrency=0).clone()
    torch.cuda.synchronize()
    s0 = rig.net.run_stats()
    assert s0["group_steps"] >= 1 and s0["small_mask_steps"] == 0          # the full-chip stream took the group launches
    # (1) a 32-CU stream (contiguous mask bits: honoured by the hardware), concurrency = 0 stated
    small = tstreams.masked_stream(range(32), "cuda:0")
    r_small = network.Runner(None, rig.net)
    with torch.cuda.stream(small):
        for _ in range(3):
            got = r_small.run_batch(x, concurrency=0)
    torch.cuda.synchronize()
    s1 = rig.net.run_stats()
    assert s1["small_mask_steps"] - s0["small_mask_steps"] == 3 and s1["group_steps"] == s0["group_steps"]
    assert bool((got == want).all())
    # (2) five threads / streams / workspaces, the library left to infer the concurrency
    n_thr, n_steps = 5, 12
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(n_thr)]
    runners = [network.Runner(None, rig.net) for _ in range(n_thr)]
    for st, rn in zip(streams, runners):                    # workspaces allocated up front (set-up, on the main thread)
        with torch.cuda.stream(st):
            rn.run_batch(x, concurrency=1)
    torch.cuda.synchronize()
    s2 = rig.net.run_stats()
    errs = []
    def work(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[i]):
                for _ in range(n_steps):
                    runners[i].run_batch(x)                 # tf2_net_run: concurrency decided by the library
        except Exception as e:                              # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    for t_ in th: t_.start()
    for t_ in th: t_.join()
    torch.cuda.synchronize()
    assert not errs, errs
    s3 = rig.net.run_stats()
    assert s3["steps"] - s2["steps"] == n_thr * n_steps
    # the very first call of the burst may still see a one-stream history; from the second stream on every step is an in-flight step
    assert s3["group_steps"] - s2["group_steps"] <= 1 and s3["inflight_steps"] - s2["inflight_steps"] >= n_thr * n_steps - 1
    for rn in runners:
        assert bool((rn._logits == want).all())
