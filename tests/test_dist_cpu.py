"""N>1 path on CPU: world_size 2, gloo.  Rank 0 packs, ONE broadcast moves the packed weight
image, rank 1 adopts it; images are sharded contiguously; no other collective is used."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from tf2_amd import config as cfg, dist as tdist, network, synth
    r, w = tdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    t = cfg.tiny_tables()
    q = synth.synth_q_values(t, 2)
    model = synth.synth_model(t, q, 2) if rank == 0 else None     # only rank 0 ever sees the weights
    net = network.NetWork(t)
    blob = tdist.broadcast_network(net, model, synth.q_text(q), device=None)
    lo, hi = tdist.shard_range(7, rank, world)
    np.save(os.path.join(out_dir, f"blob{rank}.npy"), net.packed_host())
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), np.asarray([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shard_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b0, b1 = np.load(tmp_path / "blob0.npy"), np.load(tmp_path / "blob1.npy")
    assert b0.size > 0 and (b0 == b1).all()
    s0, s1 = np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")
    assert s0.tolist() == [0, 4] and s1.tolist() == [4, 7]


def test_shard_range_covers_everything():
    from tf2_amd import dist as tdist
    for n in (1, 7, 32, 256):
        for w in (1, 2, 4, 8):
            spans = [tdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_bench_py_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a torchrun environment must start 2 ranks itself (round 1 parsed and ignored
    the flag).  --spawn-check runs exactly bench.py's launch / pin / process-group / weight-broadcast code, on gloo here."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check", "--batch", "5"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["spawn_check"] and d["n_gpus"] == 2 and d["packed_crc_all_ranks_equal"] and d["global_batch"] == 10 and d["rank0_shard"] == [0, 5]


def test_bench_py_refuses_a_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--spawn-check"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_bench_groups_table_rows_by_launch():
    """bench.py per_layer_class: rows without a launch of their own belong to the launch before them; repeats are folded."""
    sys.path.insert(0, ROOT)
    import bench
    cls = ["conv1", "1x1", "1x1", "3x3", "1x1", "1x1", "3x3", "1x1"] + ["1x1", "3x3", "1x1"] * 5 + ["pool", "fc"]
    launches = [1, 1, 0, 0, 0, 1, 1, 0] + [1] + [0] * 14 + [0, 1]
    kinds = [1] * 23 + [0, 1]
    got = bench.launch_groups(cls, launches, kinds)
    assert [n for n, _ in got] == ["conv1", "1x1+1x1+3x3+1x1", "1x1", "3x3+1x1", "5 x (1x1+3x3+1x1)", "pool", "fc"]
    assert [g for _, g in got][4] == list(range(8, 23)) and sum(len(g) for _, g in got) == len(cls)
    assert bench.launch_groups(["1x1", "1x1"], [1, 0], [1, 1]) == [("2 x (1x1)", [0, 1])]
