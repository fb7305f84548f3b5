#!/usr/bin/env python3
"""bench.py -- ResNet50 INT4w/INT8a images/s on N MI355X (BASELINE.json metric); `--net` runs the other BASELINE configurations
(SqueezeNet 1.1, VGG16, SSD300-VGG: synthetic Q values and weights) and the reference's other two shipped table programs (googlenet,
resnet50_pruned: shipped tables and Q files) through the same measurement and prints the same JSON line.

One process per GPU (torchrun contract), batches sharded with no data-path collective (weak
scaling: every rank runs `--batch` images per step); the packed weights are broadcast once
over RCCL before timing.  A "step" = one pass of the whole hot path over one batch already
resident in HBM: input quantisation + space-to-depth, all 54 layers, logits copy.  Consecutive
steps are independent batches: by default four are in flight (`--inflight`, step i on HIP stream
i % 4 with its own workspace; the HIP runtime's hardware-queue count is raised so that each of these streams has
its own queue, see GPU_MAX_HW_QUEUES below) so that the latency-bound small layers of one batch overlap with
another batch's kernels; all K timed steps complete inside the timed region, and the
one-batch-at-a-time rate is reported beside it (`images_per_s_one_batch_at_a_time`).

`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks on
127.0.0.1 (one per GPU); under torchrun it asserts WORLD_SIZE == N.  Every rank pins LOCAL_RANK -> GPU and fails
loudly if that GPU does not exist.  `--spawn-check` runs only that launch / pin / broadcast logic (gloo on CPU when
there is no GPU) -- tests/test_dist_cpu.py drives it through this file.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how `roofline` and
`cpu_baseline` are defined."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime multiplexes streams onto 4 hardware queues by default, one of them held by the null stream: a fourth
# in-flight stream would share a queue and serialise behind another batch (measured: 62.8 k img/s with 4 streams on the
# default against 74.5 k with >= 5 queues, profiles/r02_inflight_hwqueues.txt).  Read by the runtime at initialisation,
# so it is set before torch loads it; a deployment that keeps several batches in flight sets the same variable.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def layer_ops(plan):
    """algorithmic int-ops (2/MAC) and activation bytes per image, per layer."""
    rows = []
    for L in plan:
        macs = 0 if L.ipool else L.N * L.C * L.k * L.k * L.OH * L.OW
        rd = L.C * L.H * L.W
        wr = L.N * (1 if L.endpool else L.PH * L.PW)
        res = L.N * L.PH * L.PW if L.add_src >= 0 else 0
        cls = "fc" if (L.H == 1 and L.W == 1) else ("conv1" if L.src == -1 else ("1x1" if L.k == 1 else f"{L.k}x{L.k}"))
        rows.append(dict(ops=2 * macs, bytes=rd + wr + res, cls=cls))
    return rows


def launch_groups(cls, launches, kinds):
    """Table rows grouped by the launch that computes them: [(class name, [rows])].  A row without a launch of its own
    (launches == 0, kinds != 0) belongs to the launch of the rows before it (conv_bneck: 3x3 + expand; pair launches; group
    launches: a whole bottleneck, or stage 4's five).  Names: the rows' classes joined, repeats folded: '5 x (1x1+3x3+1x1)'."""
    groups = []
    for i in range(len(cls)):
        if launches[i] > 0 or not groups or kinds[i] == 0:
            groups.append([i])
        else:
            groups[-1].append(i)
    out = []
    for g in groups:
        seq = [cls[i] for i in g]
        per = next(p for p in range(1, len(seq) + 1) if len(seq) % p == 0 and seq == seq[:p] * (len(seq) // p))
        out.append(("+".join(seq[:per]) if per == len(seq) else f"{len(seq) // per} x ({'+'.join(seq[:per])})", g))
    return out


def spawn_check(args):
    """The launch path alone: process group of --gpus ranks (RCCL on GPUs, gloo on CPU), rank -> device pinning, the ONE
    collective of the data path (broadcast of the packed weights) on a small network, and a CRC agreement check."""
    import zlib
    import numpy as np
    import torch
    import torch.distributed as dist
    from tf2_amd import config as cfg, dist as tdist, network, synth
    on_gpu = torch.cuda.is_available()
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    device = None
    if on_gpu:
        if local >= torch.cuda.device_count():
            sys.exit(f"bench.py: LOCAL_RANK {local} has no GPU ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    rank, world = tdist.init_process_group()
    assert world == args.gpus
    if world > 1:
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 2)
    model = synth.synth_model(t, q, 2) if rank == 0 else None
    net = network.NetWork(t)
    tdist.broadcast_network(net, model, synth.q_text(q), device=device)
    crc = zlib.crc32(net.packed_host().tobytes()) & 0xFFFFFFFF
    crcs = [crc]
    if world > 1:
        tt = torch.tensor([crc], dtype=torch.int64, device=device if on_gpu else "cpu")
        got = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(got, tt)
        crcs = [int(g.item()) for g in got]
    lo, hi = tdist.shard_range(args.batch * world, rank, world)
    # one sharded step per rank (GPU only: the product has no CPU path): every rank runs its contiguous shard of ONE global batch
    # plus image 0 of the global batch, and the ranks compare the CRC of image 0's logits -- the same image must give the same
    # bytes on every GPU, whatever shard it rides with
    step_ok = None
    if on_gpu:
        per = max(1, min(args.batch, 4))
        gx = synth.synth_images(t, per * world, 11)
        mine = np.concatenate([gx[:1], gx[rank * per:(rank + 1) * per]])
        out = network.Runner(None, net).run_batch(torch.from_numpy(np.ascontiguousarray(mine)).to(device)).cpu().numpy()
        c0 = zlib.crc32(out[0].tobytes()) & 0xFFFFFFFF
        c0s = [c0]
        if world > 1:
            tt = torch.tensor([c0], dtype=torch.int64, device=device)
            got = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(got, tt)
            c0s = [int(g.item()) for g in got]
        step_ok = len(set(c0s)) == 1
    if rank == 0:
        print(json.dumps(dict(spawn_check=True, n_gpus=world, backend=(dist.get_backend() if world > 1 else None),
                              device=str(device) if on_gpu else "cpu", packed_crc_all_ranks_equal=len(set(crcs)) == 1,
                              sharded_step_same_logits_all_ranks=step_ok,
                              rank0_shard=[lo, hi], global_batch=args.batch * world)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--net", default="resnet50", choices=["resnet50", "squeezenet", "vgg16", "ssd300", "googlenet", "resnet50_pruned"],
                    help="network: resnet50 (the headline: shipped resnet50_Q, seeded INQ weights) or another BASELINE.json configuration "
                         "(synthetic Q values and weights); e.g. BASELINE config 4 is `--gpus 8 --net vgg16 --batch 32`; googlenet / "
                         "resnet50_pruned: the reference's other two shipped table programs with their shipped Q files (README.md:39,76-80)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region this many times (the first one is `value`, as before; all of them under `repeats` with min / "
                         "median / max): a 20-step region is 7 ms -- one scheduling hiccup is several percent")
    ap.add_argument("--mode", type=int, default=0, help="0 auto (MFMA), 1 north-star split, 2 shift only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extra-batches", type=str, default="1,64", help="also time these per-GPU batch sizes")
    ap.add_argument("--inflight", type=int, default=4,
                    help="steps (whole batches) in flight: step i runs on HIP stream i %% inflight with its own workspace; "
                         "consecutive batches are independent, so their kernels may overlap on the GPU")
    ap.add_argument("--spinup-ms", type=float, default=600.0,
                    help="untimed steps for this long before the warm-up steps: an idle MI355X sits at ~150 MHz and takes ~0.4 s of load "
                         "to reach its 2.4 GHz engine clock (tools/clock_sample.py); 0: none")
    ap.add_argument("--feeder", type=int, default=0,
                    help="1: one host thread per in-flight stream enqueues that stream's steps (tf2_amd/feeder.py); 0: one thread feeds all "
                         "(measured equal at 20 steps: the streams then start together, and steps that run in lock-step take longer)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="-1 (default): the batches-in-flight leg replays every step from a captured HIP graph (one per stream and input buffer, "
                         "captured with that leg's launch plan during the untimed steps), the one-batch-at-a-time leg launches; 1: both legs "
                         "replay graphs; 0: both launch.  Round 4, three alternating runs at 20 steps: 90.5-91.0 k (in-flight leg replayed) against "
                         "89.1-90.3 k img/s (launched); one batch at a time 60.7 k replayed against 61.2 k launched (profiles/r04_experiments.txt)")
    ap.add_argument("--buffers", type=int, default=0,
                    help="distinct input batches rotated through every timed leg (each its own device buffer; with HIP-graph replay one graph per "
                         "(stream, buffer)); 0 (default): two per stream in flight, at least 8 -- the timed region never re-reads one hot tensor")
    ap.add_argument("--stagger-layer", type=int, default=-1,
                    help=">= 0: stage-interlocked pipelining of the batches in flight -- step k+1's stream waits (hipStreamWaitEvent) for an "
                         "event that step k's run records once its layers 0..L are enqueued (tf2_net_run_ex mark_event), so a batch "
                         "enters the chip-filling first stages when its predecessor has left them; -1: off")
    ap.add_argument("--split", type=int, default=1, help="run every batch as this many sub-batches on concurrent streams (Runner.run_split)")
    ap.add_argument("--spawn-check", action="store_true", help="only launch N ranks, pin, broadcast a small network, report")
    ap.add_argument("--master-port", type=int, default=0)
    args = ap.parse_args()

    # ---- N ranks: re-exec under torchrun when started bare -------------------------------------------------------
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        port = args.master_port
        if not port:
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with --nproc-per-node {args.gpus} "
                 f"(or without torchrun, and bench.py starts the ranks itself)")
    if args.spawn_check:
        return spawn_check(args)

    import torch
    from tf2_amd import config as cfg, dist as tdist, network, synth, _lib
    import ctypes as C

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if local >= torch.cuda.device_count():
        sys.exit(f"bench.py: LOCAL_RANK {local} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)                      # before the process group: RCCL binds to the current device
    device = torch.device("cuda", local)
    rank, world = tdist.init_process_group()
    import torch.distributed as dist
    assert world == args.gpus and (world == 1 or dist.get_world_size() == args.gpus), "process group size != --gpus"

    tables, qv, seed_m, net_name, net_note = synth.bench_network(args.net)
    plan = cfg.build_plan(tables)
    qtext = synth.q_text(qv)
    model = synth.synth_model(tables, qv, seed=seed_m) if rank == 0 else None
    net = network.NetWork(tables)
    tdist.broadcast_network(net, model, qtext, device, pack_mode=args.mode)
    broadcast = dict(ms=getattr(net, "broadcast_ms", None), bytes=getattr(net, "broadcast_bytes", None),
                     note="the data path's ONE collective: the packed weight image from rank 0 (RCCL over xGMI), before any timed region")
    img_c, img_h, img_w = int(tables["INPUT_IMAGE_C"]), int(tables["INPUT_IMAGE_H"]), int(tables["INPUT_IMAGE_W"])
    runner = network.Runner(None, net)

    feeder = [None]

    def barrier():
        if feeder[0] is not None:
            feeder[0].drain()             # every step handed to the feeder threads is enqueued
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    n_inflight = max(1, args.inflight)
    # plain streams: the batches in flight SHARE every CU (rounds 2-3's "XCD-partitioned" streams never partitioned anything: DESIGN.md 3)
    fl_streams = [torch.cuda.Stream(device=device) for _ in range(n_inflight)] if n_inflight > 1 else []
    # input rotation: n_buf distinct batches; step k takes buffer k % n_buf on stream k % n_inflight (n_buf a multiple of n_inflight, so a
    # stream alternates between its own n_buf / n_inflight buffers and a captured graph is keyed by (stream, buffer))
    n_buf = args.buffers if args.buffers > 0 else max(8, 2 * n_inflight)
    n_buf = (n_buf + n_inflight - 1) // n_inflight * n_inflight
    fl_runners = [network.Runner(None, net) for _ in range(n_inflight)] if n_inflight > 1 else []
    step_no = [0]
    graphs = {}
    serial = [False]      # True: one batch at a time on the default stream (reported next to the pipelined figure)

    def one(rn, x, **kw):
        return rn.run_split(x, args.split) if args.split > 1 else rn.run_batch(x, **kw)

    stagger = args.stagger_layer >= 0 and n_inflight > 1 and args.graph != 1 and args.split == 1
    graph_inflight = args.graph == 1 or (args.graph == -1 and n_inflight > 1 and not stagger and not args.feeder)
    graph_serial = args.graph == 1
    if args.feeder and n_inflight > 1 and not graph_inflight and not stagger and args.split == 1:
        from tf2_amd.feeder import StreamFeeder
        feeder[0] = StreamFeeder(fl_streams, fl_runners, device)
    mark_ring = [torch.cuda.Event() for _ in range(2 * n_inflight)] if stagger else []
    prev_mark = [None]

    def step(xs):
        x = xs[step_no[0] % len(xs)]
        if n_inflight > 1 and not serial[0]:
            i = step_no[0] % n_inflight
            step_no[0] += 1
            if feeder[0] is not None:
                feeder[0].submit(i, lambda rn: rn.run_batch(x, concurrency=1))
                return
            with torch.cuda.stream(fl_streams[i]):
                if graph_inflight:
                    key = (i, x.data_ptr(), x.shape[0])
                    if key not in graphs:
                        graphs[key] = fl_runners[i].capture(x, split=args.split, concurrency=1)
                    graphs[key]()
                elif stagger:
                    if prev_mark[0] is not None:
                        fl_streams[i].wait_event(prev_mark[0])
                    ev = mark_ring[step_no[0] % len(mark_ring)]
                    one(fl_runners[i], x, concurrency=1, mark=(ev, args.stagger_layer))
                    prev_mark[0] = ev
                else:
                    one(fl_runners[i], x, concurrency=1)      # the caller's own statement: other batches are in flight
            return
        step_no[0] += 1
        if graph_serial:
            key = (x.data_ptr(), x.shape[0])
            if key not in graphs:
                r = network.Runner(None, net)
                graphs[key] = (r, r.capture(x, split=args.split, concurrency=0))
            graphs[key][1]()
        else:
            one(runner, x)

    input_sets = {}
    rank_dts = []

    def inputs(batch):
        """n_buf distinct synthetic batches of this size, resident in HBM (19 MB each at batch 32: together with the workspaces beyond
        what the Infinity Cache holds for one tensor; the reference reloads ONE image for every frame, runner.cpp:152-154)"""
        if batch not in input_sets:
            input_sets[batch] = [torch.from_numpy(synth.synth_images(tables, batch, seed=100 + 17 * k + rank)).to(device) for k in range(n_buf)]
        return input_sets[batch]

    def timed(batch, steps, warmup, spin=True):
        xs = inputs(batch)
        x = xs[0]
        if n_inflight > 1 and not serial[0]:   # set-up, not a step: every in-flight runner allocates its workspace ...
            for st, rn in zip(fl_streams, fl_runners):
                with torch.cuda.stream(st):
                    one(rn, x, concurrency=1)
            torch.cuda.synchronize(device)
            if graph_inflight:                 # ... and every (stream, buffer) graph the leg will replay is captured
                for k in range(n_buf):
                    i = k % n_inflight
                    key = (i, xs[k].data_ptr(), batch)
                    if key not in graphs:
                        with torch.cuda.stream(fl_streams[i]):
                            graphs[key] = fl_runners[i].capture(xs[k], split=args.split, concurrency=1)
        torch.cuda.synchronize(device)
        step_no[0] = 0
        if args.spinup_ms > 0 and spin:        # bring the device out of its idle power state (set-up, not a step)
            t_end = time.perf_counter() + args.spinup_ms * 1e-3
            while time.perf_counter() < t_end:          # time-bounded: ranks run different counts, so no collective in here
                for _ in range(4):
                    step(xs)
                if feeder[0] is not None:
                    feeder[0].drain()
                torch.cuda.synchronize(device)
        for _ in range(warmup):
            step(xs)
        barrier()
        prev_mark[0] = None                 # the timed region starts with an empty pipeline: the first step waits for nobody
        t0 = time.perf_counter()
        for _ in range(steps):
            step(xs)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            every = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(every, tt)                          # each rank's own time: the spread across GPUs is printed beside the value
            rank_dts[:] = [float(e.item()) for e in every]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, x

    # the same W + K steps once BEFORE the device is brought to its clock, reported beside the figure of record: what the
    # spin-up changes is then visible in every line this script prints
    cold = None
    if args.spinup_ms > 0:
        dc, _ = timed(args.batch, args.steps, args.warmup, spin=False)
        cold = dict(value=round(world * args.batch * args.steps / dc, 1), ms_per_step=round(dc / args.steps * 1e3, 4),
                    note="the same W warm-up + K timed steps run first, without the spin-up, on the device as the host's set-up left it "
                         "(idle power state, engine clock still ramping: tools/clock_sample.py)")
    dt, x = timed(args.batch, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    # the same region again (device warm: no further spin-up), W warm-up steps in front of each: how far the figure of record moves
    rep_vals = [value]
    for _ in range(max(0, args.repeats - 1)):
        dr, _ = timed(args.batch, args.steps, args.warmup, spin=False)
        rep_vals.append(world * args.batch * args.steps / dr)
    rs = sorted(rep_vals)
    repeats = dict(values=[round(v, 1) for v in rep_vals], min=round(rs[0], 1), median=round(float(np.median(rs)), 1), max=round(rs[-1], 1),
                   spread_pct=round(100.0 * (rs[-1] - rs[0]) / float(np.median(rs)), 2),
                   note="`value` is values[0]; the others repeat the W + K steps back to back on the warm device"
                        + ("; spread above 3 %: quote the median" if (rs[-1] - rs[0]) / float(np.median(rs)) > 0.03 else ""))
    per_rank = None
    if world > 1 and rank_dts:
        rates = [args.batch * args.steps / d for d in rank_dts]
        per_rank = dict(images_per_s_min=round(min(rates), 1), images_per_s_max=round(max(rates), 1),
                        note="each rank's own K steps / its own time; `value` = all ranks' images / the SLOWEST rank's time")

    serial_value = None
    if n_inflight > 1:
        serial[0] = True
        # >= 8 untimed steps first: the library picks its tile shapes by whether calls arrive on several streams (the last
        # eight calls, Net::run), and this leg measures the one-stream choice
        d1, _ = timed(args.batch, args.steps, max(args.warmup, 8))
        serial[0] = False
        serial_value = round(world * args.batch * args.steps / d1, 1)

    # ---- evidence for the batches-in-flight figure: GPU timestamps (HIP events on each step's own stream, against one
    #      common base event) of when every step starts and ends.  rocprofv3's kernel trace serialises the dispatches of
    #      different streams (profiles/r02_overlap_inflight3_rocprof.json: overlap factor 1.0 under the tracer), so the
    #      overlap is shown here instead: a step's latency is ~n_inflight x the interval at which steps complete.
    pipe = None
    if n_inflight > 1 and rank == 0:
        base_ev = torch.cuda.Event(enable_timing=True)
        evs = []
        torch.cuda.synchronize(device)
        base_ev.record(torch.cuda.current_stream(device))
        for st in fl_streams:
            st.wait_event(base_ev)
        n_ev = 30
        for k in range(n_ev):
            i = k % n_inflight
            with torch.cuda.stream(fl_streams[i]):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(fl_streams[i])
                one(fl_runners[i], x, concurrency=1)
                e1.record(fl_streams[i])
            evs.append((e0, e1))
        torch.cuda.synchronize(device)
        t0s = np.asarray([base_ev.elapsed_time(a) for a, _ in evs]) * 1e3
        t1s = np.asarray([base_ev.elapsed_time(b) for _, b in evs]) * 1e3
        keep = slice(n_inflight * 2, None)                   # steady state
        lat_us = float(np.mean((t1s - t0s)[keep]))
        interval_us = float((t1s[-1] - t1s[n_inflight * 2]) / (n_ev - 1 - n_inflight * 2))
        # steps whose [start, end] GPU intervals contain the end time of step k (concurrently resident batches)
        resident = float(np.mean([np.sum((t0s < t1s[k]) & (t1s >= t1s[k])) for k in range(n_inflight * 2, n_ev)]))
        pipe = dict(step_latency_us=round(lat_us, 1), completion_interval_us=round(interval_us, 1),
                    latency_over_interval=round(lat_us / interval_us, 2), steps_resident_when_one_completes=round(resident, 2),
                    note="HIP-event GPU timestamps per step on its own stream; serial execution would give latency == interval")

    sweep = {}
    for b in [int(v) for v in args.extra_batches.split(",") if v.strip()]:
        if b == args.batch:
            continue
        d2, _ = timed(b, max(3, args.steps // 2), 2)
        sweep[str(b)] = round(world * b * max(3, args.steps // 2) / d2, 1)

    # ---- batch-1 LATENCY (the reference reports "Latency ms" next to "Throughput fps", runner.cpp:187-189): one image
    #      at a time, the step replayed from a HIP graph, synchronised after every image ----
    lat = None
    if rank == 0:
        x1 = torch.from_numpy(synth.synth_images(tables, 1, seed=7)).to(device)
        r1 = network.Runner(None, net)
        res = {}
        for name, fn in (("launches", lambda: r1.run_batch(x1)), ("hip_graph", None)):
            if fn is None:
                rg = network.Runner(None, net)
                fn = rg.capture(x1, concurrency=0)
            for _ in range(5):
                fn()
            torch.cuda.synchronize(device)
            n_lat = 50
            t0 = time.perf_counter()
            for _ in range(n_lat):
                fn()
                torch.cuda.synchronize(device)
            res[name] = round((time.perf_counter() - t0) / n_lat * 1e6, 1)
        lat = dict(us_per_image=min(res.values()), by_path=res, note="batch 1, one image at a time, host-synchronised after each")

    # ---- roofline: live per-layer HIP-event timing on the launch stream (C-side hook), for BOTH launch plans ----
    lo = layer_ops(plan)
    PEAK_I8 = 5000.0    # TOP/s dense int8 MFMA (MI355X_MICROARCH.md: ~2x the 2.5 PF bf16 dense peak)
    PEAK_HBM = 8000.0   # GB/s
    prof_steps = 5

    def profile_plan(conc):
        """One batch at a time on one stream with the launch plan of `conc` (0: the one-batch-at-a-time plan, 1: the plan the
        timed region runs with batches in flight): per-launch HIP-event times, classes by launch, kernel names from the
        library's own launch list."""
        for _ in range(3):
            runner.run_batch(x, concurrency=conc)
        _lib.check(_lib.lib().tf2_net_profile(net._h, 1))
        for _ in range(prof_steps):
            runner.run_batch(x, concurrency=conc)
        torch.cuda.synchronize(device)
        ms = np.zeros(len(plan), np.float32); nl = np.zeros(len(plan), np.int32); kinds = np.zeros(len(plan), np.int32)
        _lib.check(_lib.lib().tf2_net_profile_read(net._h, ms.ctypes.data, nl.ctypes.data, kinds.ctypes.data, len(plan)))
        # one event pair around the whole layer loop: what the per-layer pairs add by themselves (each record is a
        # marker the command processor handles between kernels) is removed by rescaling the per-layer sum to it
        _lib.check(_lib.lib().tf2_net_profile(net._h, 2))
        for _ in range(prof_steps):
            runner.run_batch(x, concurrency=conc)
        torch.cuda.synchronize(device)
        loop_ms, loop_n = C.c_float(0), C.c_int32(0)
        _lib.check(_lib.lib().tf2_net_profile_loop_read(net._h, C.byref(loop_ms), C.byref(loop_n)))
        _lib.check(_lib.lib().tf2_net_profile(net._h, 0))
        per_layer_ms = ms / np.maximum(nl, 1)
        event_scale = 1.0
        if loop_n.value > 0 and per_layer_ms.sum() > 0:
            event_scale = min(1.0, (loop_ms.value / loop_n.value) / float(per_layer_ms.sum()))
        per_layer_ms = per_layer_ms * event_scale
        launches = net.describe_launches(args.batch, conc)
        kern_of = {}
        for l in launches:
            if l["layer"] >= 0:
                kern_of.setdefault(l["layer"], []).append(l["kernel"].split("<")[0].split(" ")[0])
        # classes by LAUNCH: a launch that computes several table rows (conv_bneck: 3x3 + expand; pair launches; group / band launches:
        # a whole bottleneck, or several) is its own class, named after the rows it covers -- its time cannot be split between them
        classes = {}
        for name, g in launch_groups([r["cls"] for r in lo], nl, kinds):
            c = classes.setdefault(name, dict(ops=0, bytes=0, ms=0.0, kernel=set(), launches=0))
            c["launches"] += 1
            c["kernel"].update(kern_of.get(g[0], ["none"]))
            for i in g:
                c["ops"] += lo[i]["ops"] * args.batch; c["bytes"] += lo[i]["bytes"] * args.batch
                c["ms"] += float(per_layer_ms[i])
        per_class = {k: dict(kernel="+".join(sorted(v["kernel"])), launches=v["launches"], ms=round(v["ms"], 4),
                             tops=round(v["ops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                             frac_int8_peak=round(v["ops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_I8, 4) if v["ms"] > 0 else None,
                             gbps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None,
                             frac_hbm_peak=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM, 4) if v["ms"] > 0 else None)
                     for k, v in classes.items()}
        # the literal shift-accumulate kernel (modes 1 / 2: the north-star split) is priced against ITS roof: one v_mad_i32_i24 per MAC and lane,
        # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz x ... = 78 TMAC/s (conv_shift.hip)
        for k, v in classes.items():
            if any("conv_shift" in kk for kk in v["kernel"]) and v["ms"] > 0:
                per_class[k]["tmacs"] = round(v["ops"] / 2 / (v["ms"] * 1e-3) / 1e12, 2)
                per_class[k]["frac_valu_shift_peak"] = round(v["ops"] / 2 / (v["ms"] * 1e-3) / 1e12 / 78.0, 4)
        cv = [i for i in range(len(plan)) if kinds[i] in (1, 2)]
        n_launch = max(1, sum(1 for i in cv if nl[i] > 0))
        dom_ms = float(sum(per_layer_ms[i] for i in cv))
        # the launch that takes the most time: the plan's dominant kernel
        top = max((i for i in cv if nl[i] > 0), key=lambda i: per_layer_ms[i], default=None)
        top_kernel = None
        if top is not None:
            full = [l["kernel"] for l in launches if l["layer"] == top]
            top_kernel = dict(kernel=full[0] if full else None, first_row=int(top), us=round(float(per_layer_ms[top]) * 1e3, 2))
        return dict(per_layer_ms=per_layer_ms, nl=nl, kinds=kinds, cv=cv, n_launch=n_launch, dom_ms=dom_ms, event_scale=event_scale,
                    per_class=per_class, top_kernel=top_kernel, kernels=sorted({k for v in kern_of.values() for k in v if k.startswith("conv")}))

    ROUND = "r06"

    def committed(name):
        """a profile of THIS round committed under profiles/ for this network / batch / kernel mode (tools/round_evidence.sh)"""
        pj = os.path.join(ROOT, "profiles", name)
        return pj if (os.path.exists(pj) and args.mode == 0) else None

    tag = "" if args.net == "resnet50" else f"{args.net}_"

    def rocprof_of(name):
        """the committed rocprofv3 --kernel-trace summary of the same workload (ONLY batch-N steps in the profiled process, one per
        launch plan): the conv kernels' time per step as the profiler sees it, next to the live figure"""
        pj = committed(name)
        if not pj:
            return None, None
        rs = json.load(open(pj))
        return round(rs["conv_us_per_step"], 1), os.path.basename(pj)

    def traffic_of(name, P):
        """mean HBM bytes per conv launch from the committed PMC passes of that plan (FETCH_SIZE doubled per the guide's gfx950 note,
        WRITE_SIZE; separate --pmc runs)"""
        pj = committed(name)
        if not pj:
            return None, "no PMC pass committed for this network / batch size / kernel mode"
        pm = json.load(open(pj))
        if len(pm.get("layers", [])) != len(plan):
            return None, f"{os.path.basename(pj)} does not describe this table program"
        tb = sum(pm["layers"][i]["fetch_bytes"] + pm["layers"][i]["write_bytes"] for i in P["cv"])
        return round(tb / P["n_launch"]), (f"mean HBM bytes per launch over the step's {P['n_launch']} conv launches, rocprofv3 FETCH_SIZE(x2, gfx950)+WRITE_SIZE "
                                           f"in separate --pmc passes, {os.path.basename(pj)}")

    P0 = profile_plan(0)
    P1 = profile_plan(1) if n_inflight > 1 else None
    cv = P0["cv"]
    dom_ops = sum(lo[i]["ops"] for i in cv) * args.batch
    alg_bytes = sum(lo[i]["bytes"] for i in cv) * args.batch

    def plan_block(P, conc, note):
        """one launch plan, its launches timed ONE BATCH AT A TIME on one stream: algorithmic bytes (ops) / sum of HIP-event durations"""
        us, src = rocprof_of(f"{ROUND}_rocprof_{tag}b{args.batch}{'_conc1' if conc else ''}_summary.json")
        tr, tr_note = traffic_of(f"{ROUND}_pmc_conv_{tag}b{args.batch}{'_conc1' if conc else ''}.json", P)
        g = alg_bytes / (P["dom_ms"] * 1e-3) / 1e9 if P["dom_ms"] > 0 else 0.0
        tp = dom_ops / (P["dom_ms"] * 1e-3) / 1e12 if P["dom_ms"] > 0 else 0.0
        k = P["top_kernel"]
        dom = None
        if k is not None:
            # the dominant launch against ITS OWN roofs: the algorithmic bytes / ops of the table rows it computes over its own duration
            rows = next(g_ for _, g_ in launch_groups([r["cls"] for r in lo], P["nl"], P["kinds"]) if g_[0] == k["first_row"])
            kb = sum(lo[i]["bytes"] for i in rows) * args.batch
            ko = sum(lo[i]["ops"] for i in rows) * args.batch
            dom = dict(k, rows=[int(r) for r in rows], algorithmic_bytes=kb, achieved_gbps=round(kb / (k["us"] * 1e-6) / 1e9, 1),
                       frac_hbm_peak=round(kb / (k["us"] * 1e-6) / 1e9 / PEAK_HBM, 4), achieved_tops=round(ko / (k["us"] * 1e-6) / 1e12, 1),
                       frac_int8_peak=round(ko / (k["us"] * 1e-6) / 1e12 / PEAK_I8, 4))
        return dict(kernel=dom, kernels=P["kernels"], launches_per_step=P["n_launch"], algorithmic_bytes_per_launch=round(alg_bytes / P["n_launch"]),
                    avg_launch_us=round(P["dom_ms"] / P["n_launch"] * 1e3, 2), kernel_us_per_step=round(P["dom_ms"] * 1e3, 1),
                    kernel_us_per_step_rocprof=us, rocprof_summary=src, event_pair_scale=round(P["event_scale"], 4),
                    one_at_a_time=dict(achieved=round(g, 1), frac=round(g / PEAK_HBM, 4), mfma_tops=round(tp, 1), mfma_frac=round(tp / PEAK_I8, 4)),
                    traffic=tr, traffic_note=tr_note, note=note)

    note0 = ("the one-batch-at-a-time plan (group launches): algorithmic bytes (activations read + written + residual read; SURVEY.md 8(d)) of the "
             "step's conv launches / sum of their HIP-event durations on the launch stream; `kernel` = the launch that takes the most time, named "
             "by tf2_net_describe_launches, against its own rows' bytes and ops; per-layer event times rescaled by event_pair_scale = (one event "
             "pair around the whole layer loop) / (their sum)")
    one_batch = plan_block(P0, 0, note0)
    total_bytes = sum(r["bytes"] for r in lo) * args.batch
    hbm_gbps = total_bytes / (ms_per_step * 1e-3) / 1e9
    # TOP LEVEL = the plan the timed region runs (round 5; rounds 1-4 had the one-batch plan here and this one nested under `in_flight`):
    # achieved = the conv launches' algorithmic bytes per step / ms_per_step of the TIMED region = the rate the pipeline sustains with
    # batches in flight; the same launches one at a time are `one_at_a_time` (the profiler serialises streams, so their overlap cannot be
    # traced: pipeline_evidence), the other plan is `one_batch`
    if P1 is not None:
        roofline = plan_block(P1, 1, "the plan of the TIMED region (batches in flight: band launches, no group launches): achieved / frac = the conv launches' "
                                     "algorithmic bytes per step / ms_per_step; kernel_us_per_step* and one_at_a_time: the SAME launches run one batch at a "
                                     "time on one stream; overlap_factor = kernel time / step time")
        ach = alg_bytes / (ms_per_step * 1e-3) / 1e9
        roofline = dict(dict(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM, unit="GB/s", frac=round(ach / PEAK_HBM, 4)), **roofline)
        roofline["mfma_side"] = dict(achieved_tops=round(dom_ops / (ms_per_step * 1e-3) / 1e12, 1), peak_tops=PEAK_I8,
                                     frac=round(dom_ops / (ms_per_step * 1e-3) / 1e12 / PEAK_I8, 4))
        roofline["overlap_factor"] = round(P1["dom_ms"] / ms_per_step, 2) if ms_per_step > 0 else None
        roofline["one_batch"] = one_batch
    else:
        roofline = dict(dict(bound="hbm", achieved=one_batch["one_at_a_time"]["achieved"], peak=PEAK_HBM, unit="GB/s", frac=one_batch["one_at_a_time"]["frac"]), **one_batch)
        roofline["mfma_side"] = dict(achieved_tops=one_batch["one_at_a_time"]["mfma_tops"], peak_tops=PEAK_I8, frac=one_batch["one_at_a_time"]["mfma_frac"])
    # per layer class (north_star: "%-of-int8-roofline reported per layer class"), keyed by LAUNCH; top level = the timed plan's classes,
    # each with its share of that plan's kernel time (a class's sustained rate in the timed region = its one-at-a-time rate x overlap_factor)
    Pt = P1 if P1 is not None else P0
    tot1 = sum(v["ms"] for v in Pt["per_class"].values()) or 1.0
    per_class = {k: dict(v, share_of_kernel_time=round(v["ms"] / tot1, 4)) for k, v in Pt["per_class"].items()}
    per_class_one_batch = P0["per_class"] if P1 is not None else None

    # ---- CPU baseline: the oracle (restated reference CPU path) on the host cores, rank 0, N=1 ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import netref, oracle as O
        ref = netref.RefNet(tables, qv, model)
        n_done, t_cpu = 0, 0.0
        imgs = synth.synth_images(tables, 64 if args.net == "resnet50" else 16, seed=100)
        chunk = max(1, min(16, O.num_threads()))          # tf2o_layer parallelises over (image, output channel): every thread busy
        first_logits = None
        while t_cpu < args.cpu_seconds and n_done + chunk <= len(imgs):
            t0 = time.perf_counter()
            outs = ref.run(imgs[n_done:n_done + chunk])
            t_cpu += time.perf_counter() - t0
            if first_logits is None:
                first_logits = ref.logits(outs)
            n_done += chunk
        # parity gate: the GPU logits of the same images equal the oracle's
        got = runner.run_batch(torch.from_numpy(imgs[:first_logits.shape[0]]).to(device)).cpu().numpy()
        parity = bool((got == first_logits).all())
        cpu = dict(value=round(n_done / t_cpu, 3), unit="images/s", cores=O.num_threads(), kind="port",
                   sample=f"{n_done} synthetic {img_h}x{img_w} images, same {net_name} weights/Q, oracle/tf2_oracle.c (OpenMP over image x output "
                          f"channel, {O.num_threads()} threads = the CPUs this process may use: affinity {len(os.sched_getaffinity(0))}, "
                          f"cgroup quota applied; {os.cpu_count()} logical CPUs visible) in {t_cpu:.1f} s",
                   parity_with_gpu_logits=parity)

    if rank == 0:
        line = dict(metric=f"images/sec {net_name} INT4w/INT8a", value=round(value, 1), unit="images/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4), higher_is_better=True,
                    scaling="weak", vs_baseline=None, dtype="int8", data="synthetic",
                    config=dict(workload=f"{net_name} INT4w/INT8a ({net_note}), "
                                         f"batch {args.batch}/GPU, {img_c}x{img_h}x{img_w} float images resident in HBM",
                                global_batch=args.batch * world, parallelism=f"dp{world}", kernel_mode=args.mode,
                                sub_batches_per_step=args.split, batches_in_flight=n_inflight, hip_graph=dict(in_flight_leg=bool(graph_inflight), one_batch_at_a_time_leg=bool(graph_serial)),
                                stage_interlock_layer=(args.stagger_layer if stagger else None),
                                input_buffers_rotated=n_buf,
                                host_feeder_threads=(n_inflight if feeder[0] is not None else 1),
                                spinup_ms=args.spinup_ms,
                                spinup_note="untimed steps for spinup_ms before the W warm-up steps of every timed leg: an idle MI355X sits "
                                            "at ~150 MHz and needs ~0.4 s of load to reach 2.4 GHz (tools/clock_sample.py); the timed region is "
                                            "still exactly K steps between barrier + synchronize"),
                    repeats=repeats, cold_start=cold, weight_broadcast=broadcast, per_rank=per_rank, roofline=roofline, cpu_baseline=cpu,
                    hbm=dict(algorithmic_gbps=round(hbm_gbps, 1), frac_of_8tbps=round(hbm_gbps / PEAK_HBM, 4),
                             bytes_per_image=sum(r["bytes"] for r in lo)),
                    per_layer_class=per_class, per_layer_class_one_batch=per_class_one_batch, images_per_s_by_batch=sweep,
                    images_per_s_one_batch_at_a_time=serial_value, latency_batch1=lat, pipeline_evidence=pipe)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
