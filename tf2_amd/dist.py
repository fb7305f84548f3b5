"""Multi-GPU layout of the cnn path: one process per GPU, batches shard naturally
(images are independent, no layer mixes batch elements), ONE collective -- the one-time
broadcast of the packed weight image from rank 0 (RCCL over xGMI; `nccl` backend IS RCCL on
ROCm) -- and no cross-GPU reduction.  The reference has no multi-device inference path at all
(opencl_fpga.cpp:42-43 uses devices[0])."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import network


def init_process_group(backend: Optional[str] = None):
    """Reads RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of `n_items` images for `rank` (B/G per rank, remainder to
    the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_network(net: network.NetWork, model, q_file, device, pack_mode: int = 0, src: int = 0):
    """Rank `src` encodes + packs (Quantization, LoadModel, Pack); every rank receives the
    packed image by ONE broadcast and binds it.  All ranks parse the (tiny) Q file themselves.
    Works with the gloo backend on CPU tensors too (tests): then nothing is bound."""
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if distributed else 0
    net.Quantization(q_file)
    if rank == src:
        net.LoadModel(model)
        net.Pack(pack_mode)
        blob = torch.from_numpy(net.packed_host())
        size = torch.tensor([blob.numel()], dtype=torch.int64)
    else:
        blob, size = None, torch.tensor([0], dtype=torch.int64)
    on_gpu = device is not None and str(device) != "cpu"
    net.broadcast_ms, net.broadcast_bytes = None, int(size.item()) if rank == src else None
    if distributed:
        import time
        size_d = size.to(device) if on_gpu else size
        dist.broadcast(size_d, src=src)
        n = int(size_d.item())
        if rank == src:
            blob_d = blob.to(device) if on_gpu else blob
        else:
            blob_d = torch.empty(n, dtype=torch.uint8, device=device if on_gpu else "cpu")
        if on_gpu:
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        dist.broadcast(blob_d, src=src)          # the one collective of the data path
        if on_gpu:
            torch.cuda.synchronize(device)
        net.broadcast_ms, net.broadcast_bytes = round((time.perf_counter() - t0) * 1e3, 3), n
        if rank != src:
            net.adopt_packed(blob_d.cpu().numpy())
    else:
        blob_d = blob.to(device) if on_gpu else blob
    if on_gpu:
        net.InitBuffer(device, packed_dev=blob_d)
    return blob_d
