"""Host-side mirror of the reference's NetWork / Runner / network_helper interface.

Reference (Runtime_Engine/cnn/host): ``NetWork::Init(platform, model_file, q_file,
image_file, num_images)`` (inc/network.h:29-66, src/network.cpp:22-150), ``Runner::Init`` /
``Runner::Run`` (inc/runner.h:21-38, src/runner.cpp:54-198), ``Verify`` / ``Evaluation``
(src/network_helper.cpp:18-207).  Same names, argument meaning and data contract (int8
logits + top-5 labels); errors are exceptions instead of ``exit()``.

PyTorch-ROCm is used only as plumbing: device allocations, the current HIP stream and
``torch.distributed``.  All arithmetic happens in the C-ABI library (tf2_amd/_lib.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _lib, config as cfg


def _layer_descs(tables: cfg.NetTables):
    plan = cfg.build_plan(tables)
    arr = (_lib.LayerDesc * len(plan))()
    for i, L in enumerate(plan):
        d = arr[i]
        d.src, d.q_in_row, d.C, d.H, d.W = L.src, L.q_in_row, L.C, L.H, L.W
        d.N, d.k, d.stride, d.pad_h, d.pad_w, d.dil = L.N, L.k, L.stride, L.pad_h, L.pad_w, L.dil
        d.OH, d.OW = L.OH, L.OW
        d.bias_en, d.bn_en, d.relu, d.ipool = L.bias_en, L.bn_en, L.relu, L.ipool
        d.pool_en, d.pool_S, d.pool_st, d.pool_pad, d.PH, d.PW = L.pool_en, L.pool_S, L.pool_st, L.pool_pad, L.PH, L.PW
        d.add_src, d.add_relu, d.endpool, d.endpool_mult = L.add_src, L.add_relu, L.endpool, L.endpool_mult
        d.concat, d.n_start, d.model_C, d.model_k = L.concat, L.n_start, L.model_C, L.model_k
    nd = _lib.NetDesc()
    nd.n_layers = len(plan)
    nd.n_conv = int(tables["NUM_CONVOLUTIONS"])
    nd.n_q_rows = int(tables["NUM_Q_LAYERS"])
    nd.max_out_channel = int(tables["MAX_OUT_CHANNEL"])
    nd.image_c, nd.image_h, nd.image_w = int(tables["INPUT_IMAGE_C"]), int(tables["INPUT_IMAGE_H"]), int(tables["INPUT_IMAGE_W"])
    nd.conv1_rewrite = int(tables.get("xConv1Rewrite", 0))
    nd.n_concat = max([L.concat for L in plan] + [-(L.src + 2) for L in plan if L.src <= -2] + [-1]) + 1
    return plan, nd, arr


class NetWork:
    """network.h:29-66.  ``tables`` replaces the compile-time ``-DRESNET50`` header choice
    (cnn.h:29-35): pass NetTables (from config.parse_net_header / tables_from_fpganetwork /
    a builder) or a path to a ``<net>.h`` / ``fpganetwork.bin``."""

    def __init__(self, tables, netname: str = "resnet50"):
        if isinstance(tables, (str, os.PathLike)):
            p = os.fspath(tables)
            tables = cfg.tables_from_fpganetwork(cfg.read_fpganetwork(p), netname) if p.endswith(".bin") \
                else cfg.parse_net_header(p)
        self.tables: cfg.NetTables = tables
        self.plan, self._nd, self._descs = _layer_descs(tables)
        self.num_layer = len(self.plan)
        h = C.c_void_p()
        _lib.check(_lib.lib().tf2_net_create(C.byref(self._nd), self._descs, C.byref(h)))
        self._h = h
        self.q: Optional[np.ndarray] = None          # runtime q table (negated), [NUM_Q_LAYERS][MAX_OUT_CHANNEL]
        self.model_file = self.q_file = self.image_file = None
        self.num_images = 1
        self.output: Optional[np.ndarray] = None     # int8 logits [num_images][N_last]
        self.top_labels: List[List[int]] = []
        self._packed_dev = None                      # torch uint8 tensor holding the packed image

    # -- lifetime ---------------------------------------------------------------------
    def CleanUp(self):
        if getattr(self, "_h", None):
            _lib.lib().tf2_net_destroy(self._h)
            self._h = None
        self._packed_dev = None

    def __del__(self):
        try:
            self.CleanUp()
        except Exception:
            pass

    # -- Init = InitNetwork (Quantization + LoadModel) + InitBuffer ----------------------
    def Init(self, model_file, q_file, image_file=None, num_images: int = 1, device=None, pack_mode: int = 0):
        """network.cpp:22-38.  Files may be paths or in-memory arrays (model: float32 array,
        q: text/bytes or int sequence)."""
        self.model_file, self.q_file, self.image_file, self.num_images = model_file, q_file, image_file, num_images
        self.Quantization(q_file)
        self.LoadModel(model_file)
        self.Pack(pack_mode)
        if device is not None:
            self.InitBuffer(device)
        return True

    def Quantization(self, q_file):
        """quantization.cpp:25-55."""
        if isinstance(q_file, (str, os.PathLike)):
            with open(q_file, "rb") as f:
                text = f.read()
        elif isinstance(q_file, (bytes, bytearray)):
            text = bytes(q_file)
        else:
            text = ("\n".join(str(int(v)) for v in q_file) + "\n").encode()
        rows, maxc = self._nd.n_q_rows, self._nd.max_out_channel
        q = np.zeros((rows, maxc), np.int8)
        nread = C.c_int32(0)
        _lib.check(_lib.lib().tf2_quantization(self._h, text, len(text), q.ctypes.data, q.size, C.byref(nread)))
        self.q = q
        self.q_values_read = nread.value
        _lib.check(_lib.lib().tf2_net_set_q(self._h, q.ctypes.data, q.size))
        return q

    def LoadModel(self, model_file):
        """model_loader.cpp:129-258 (float32 stream -> byte codes + BiasBnParam)."""
        if isinstance(model_file, (str, os.PathLike)):
            model = np.fromfile(model_file, dtype=np.float32)
        else:
            model = np.ascontiguousarray(model_file, np.float32).ravel()
        _lib.check(_lib.lib().tf2_net_load_model(self._h, model.ctypes.data, model.size))

    def LoadModel4bit(self, model_file):
        """The same from TransForm_Kit's 4-bit packed model file (Compression/compress_net/4bit_data_format.txt;
        path or bytes) -- decoded inside the library, bit-identical to loading the float32 file it was made from."""
        if isinstance(model_file, (str, os.PathLike)):
            with open(model_file, "rb") as f:
                data = f.read()
        else:
            data = bytes(model_file)
        buf = np.frombuffer(data, np.uint8)
        _lib.check(_lib.lib().tf2_net_load_model_4bit(self._h, buf.ctypes.data, buf.size))

    def codes(self, layer: int) -> np.ndarray:
        L = self.plan[layer]
        n = C.c_size_t(0)
        _lib.check(_lib.lib().tf2_net_get_codes(self._h, layer, None, 0, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        _lib.check(_lib.lib().tf2_net_get_codes(self._h, layer, buf.ctypes.data, buf.size, C.byref(n)))
        return buf.reshape(L.N, L.C, L.k, L.k)

    def bias_bn(self, layer: int):
        N = self.plan[layer].N
        b, a, be = (np.empty(N, np.int32) for _ in range(3))
        _lib.check(_lib.lib().tf2_net_get_bias_bn(self._h, layer, b.ctypes.data, a.ctypes.data, be.ctypes.data, N))
        return b, a, be

    # -- packed weights -------------------------------------------------------------------
    def Pack(self, mode: int = 0):
        _lib.check(_lib.lib().tf2_net_pack(self._h, mode))

    def packed_host(self) -> np.ndarray:
        n = _lib.lib().tf2_net_packed_size(self._h)
        buf = np.empty(n, np.uint8)
        _lib.check(_lib.lib().tf2_net_packed_copy(self._h, buf.ctypes.data, n))
        return buf

    def adopt_packed(self, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, np.uint8)
        _lib.check(_lib.lib().tf2_net_packed_adopt(self._h, blob.ctypes.data, blob.size))

    def InitBuffer(self, device="cuda:0", packed_dev=None):
        """network.cpp:100-150: upload the weights.  ``packed_dev``: an already-resident torch
        uint8 tensor (e.g. received by the RCCL broadcast in tf2_amd.dist)."""
        import torch
        if packed_dev is None:
            packed_dev = torch.from_numpy(self.packed_host()).to(device)
        assert packed_dev.dtype == torch.uint8 and packed_dev.is_contiguous()
        self._packed_dev = packed_dev
        _lib.check(_lib.lib().tf2_net_bind_device(self._h, packed_dev.data_ptr(), packed_dev.numel()))
        self.device = packed_dev.device

    def reload_options(self):
        """Re-read the option string TF2_AMD_OPTS (csrc/opts.h) from the environment (sampled at creation otherwise)."""
        _lib.check(_lib.lib().tf2_net_reload_options(self._h))

    def describe_launches(self, batch: int, concurrency: int = 0):
        """The kernel launches of one step, as the library's launch plan selects them: list of dicts
        {layer, kernel, grid, block, lds_bytes, vgprs} (tf2_net_describe_launches; no device needed)."""
        import ctypes as C
        rows = (_lib.LaunchInfo * 512)()
        n = C.c_int(0)
        _lib.check(_lib.lib().tf2_net_describe_launches(self._h, batch, concurrency, rows, 512, C.byref(n)))
        return [dict(layer=r.layer, kernel=r.kernel.decode(), grid=r.grid, block=r.block, lds_bytes=r.lds_bytes, vgprs=r.vgprs)
                for r in rows[:n.value]]

    def run_stats(self):
        """{steps, group_steps, inflight_steps, small_mask_steps} since the handle was created (tf2_net_run_stats)."""
        import numpy as _np
        out = _np.zeros(4, _np.int64)
        _lib.check(_lib.lib().tf2_net_run_stats(self._h, out.ctypes.data))
        return dict(steps=int(out[0]), group_steps=int(out[1]), inflight_steps=int(out[2]), small_mask_steps=int(out[3]))

    def describe_workspace(self, batch: int, keep_all: bool = False):
        """(tensors, rows) of the liveness-planned workspace: tensors = [{offset, bytes, first_row, last_row}], rows = per table row
        {in_tensor, out_tensor, conv_tensor, res_tensor} (tf2_net_describe_workspace; no device needed)."""
        import ctypes as C
        nl = len(self.plan)
        ts = (_lib.TensorInfo * 1024)()
        rs = (_lib.RowTensors * nl)()
        n = C.c_int(0)
        _lib.check(_lib.lib().tf2_net_describe_workspace(self._h, batch, int(keep_all), ts, 1024, C.byref(n), rs, nl))
        return ([dict(offset=t.offset, bytes=t.bytes, first_row=t.first_row, last_row=t.last_row) for t in ts[:n.value]],
                [dict(in_tensor=r.in_tensor, out_tensor=r.out_tensor, conv_tensor=r.conv_tensor, res_tensor=r.res_tensor) for r in rs])

    def workspace_size(self, batch: int, keep_all: bool = False) -> int:
        return int(_lib.lib().tf2_net_workspace_size(self._h, batch, int(keep_all)))


class Runner:
    """runner.h:21-38.  ``Run`` executes ``num_images`` frames; unlike the reference (which
    re-reads the same image file for every frame, runner.cpp:152-154) a batch of distinct
    images can be supplied."""

    def __init__(self, platform, network: NetWork):
        self.platform = platform            # kept for signature parity; unused (no OpenCL platform)
        self.network = network
        self._ws = None
        self._ws_batch = (0, False)
        self._logits = None

    def Init(self):
        self.image_file = self.network.image_file
        self.num_images = self.network.num_images

    def _ensure(self, batch: int, keep_all: bool):
        import torch
        net = self.network
        if self._ws is None or self._ws_batch != (batch, keep_all):
            self._split = None
            size = net.workspace_size(batch, keep_all)
            self._ws = torch.empty(max(size, 256), dtype=torch.uint8, device=net.device)
            self._ws_batch = (batch, keep_all)
            nbytes = int(_lib.lib().tf2_net_logits_size(net._h, batch))       # [batch][H_last*W_last][N_last]
            N = net.plan[-1].N
            self._logits = torch.empty((batch, N) if nbytes == batch * N else (batch, nbytes // (batch * N), N),
                                       dtype=torch.int8, device=net.device)

    def run_batch(self, images, keep_all: bool = False, concurrency: int = -1, mark=None):
        """images: torch tensor on the network's device, float32 [B,C,H,W] (preprocessed
        floats as in the image .bin files) or int8 (already quantised).  Returns the int8
        logits tensor [B, N_last] (device).  Enqueued on the current HIP stream.
        concurrency: -1 let the library decide from its stream history, 0 this batch runs alone, 1 other batches are in
        flight on other streams (tile-shape choice only; same results).  mark = (torch.cuda.Event, layer): the event is
        recorded on the stream once the launches of layers 0..layer are enqueued (tf2_net_run_ex)."""
        import torch
        import ctypes as C
        net = self.network
        assert images.is_contiguous() and images.device == net.device
        B = images.shape[0]
        self._ensure(B, keep_all)
        stream = torch.cuda.current_stream(net.device).cuda_stream
        assert images.dtype in (torch.float32, torch.int8)
        if concurrency == -1 and mark is None:
            fn = _lib.lib().tf2_net_run if images.dtype == torch.float32 else _lib.lib().tf2_net_run_q
            _lib.check(fn(net._h, images.data_ptr(), B, self._ws.data_ptr(), self._ws.numel(),
                          self._logits.data_ptr(), stream))
            return self._logits
        o = _lib.RunOpts(C.sizeof(_lib.RunOpts), int(images.dtype == torch.int8), int(concurrency), -1, None)
        if mark is not None:
            ev, layer = mark
            ev.record(torch.cuda.current_stream(net.device))       # creates the underlying hipEvent_t; re-recorded by the library
            o.mark_event = ev.cuda_event
            o.mark_after_layer = int(layer)
        _lib.check(_lib.lib().tf2_net_run_ex(net._h, images.data_ptr(), B, self._ws.data_ptr(), self._ws.numel(),
                                             self._logits.data_ptr(), stream, C.byref(o)))
        return self._logits

    def poll_error(self, batch: int = None):
        """Did a group launch of a step on this runner's workspace give up a meeting (tf2_net_poll_error)?  Synchronises the current
        stream; raises Tf2Error (status TF2_ERR_GROUP = -6) once per report, else returns None.  batch: the batch the workspace was last
        used for (default: the runner's current one)."""
        import torch
        net = self.network
        stream = torch.cuda.current_stream(net.device).cuda_stream
        _lib.check(_lib.lib().tf2_net_poll_error(net._h, int(batch or self._ws_batch[0]), self._ws.data_ptr(), self._ws.numel(), stream))

    def run_split(self, images, parts: int = 2):
        """One batch as `parts` sub-batches on concurrent HIP streams (fork from / join to the current stream): the
        small-map layers of a batch are bound by their own launch-to-drain latency, not by throughput, so two halves
        interleave on the GPU.  Same logits tensor [B, N_last] as run_batch; images are independent, so the result
        is bit-identical (tests/test_gpu_configs.py)."""
        import torch
        net = self.network
        B = images.shape[0]
        parts = max(1, min(parts, B))
        if parts == 1:
            return self.run_batch(images)
        if getattr(self, "_split", None) is None or self._split[0] != (B, parts):
            bounds = [((B * i) // parts, (B * (i + 1)) // parts) for i in range(parts)]
            nbytes = int(_lib.lib().tf2_net_logits_size(net._h, B)); N = net.plan[-1].N
            logits = torch.empty((B, N) if nbytes == B * N else (B, nbytes // (B * N), N), dtype=torch.int8, device=net.device)
            subs = []
            for lo, hi in bounds:
                r = Runner(None, net)
                r._ensure(hi - lo, False)
                r._logits = logits[lo:hi]                  # contiguous rows of the shared output
                subs.append((r, torch.cuda.Stream(device=net.device), lo, hi))
            self._split = ((B, parts), subs, logits)
        _, subs, logits = self._split
        cur = torch.cuda.current_stream(net.device)
        fork = torch.cuda.Event(); fork.record(cur)
        for r, st, lo, hi in subs:
            st.wait_event(fork)
            with torch.cuda.stream(st):
                r.run_batch(images[lo:hi])
                join = torch.cuda.Event(); join.record(st)
            cur.wait_event(join)
        self._logits = logits
        return logits

    def capture(self, images, split: int = 1, concurrency: int = -1):
        """Capture one run_batch(images) into a HIP graph (launch-bound small batches: the ~57
        launches of a step replay as one graph launch).  `images` is a static input buffer: refill
        it in place, call the returned function, read `self._logits`.
        concurrency as in run_batch -- state it: the capture runs on a side stream, so the library's stream history (-1) takes
        the step for one of several batches in flight and picks that launch plan (no group launches)."""
        import torch
        net = self.network
        step = (lambda: self.run_split(images, split)) if split > 1 else (lambda: self.run_batch(images, concurrency=concurrency))
        step()                                      # warm-up: lazy attribute setup must not be captured
        torch.cuda.synchronize(net.device)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=net.device)
        side.wait_stream(torch.cuda.current_stream(net.device))
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                step()
        torch.cuda.current_stream(net.device).wait_stream(side)
        self._graph = g
        return g.replay

    def read_layer(self, layer: int, batch: int) -> np.ndarray:
        """Output of ``layer`` after a keep_all run, as NCHW int8 (per-layer parity tests)."""
        import torch
        net = self.network
        L = net.plan[layer] if layer >= 0 else None
        if layer < 0:
            shape = (batch, net.plan[0].C, net.plan[0].H, net.plan[0].W)
        elif L.endpool:
            shape = (batch, L.N, 1, 1)
        else:
            shape = (batch, L.N, L.PH, L.PW)
        out = np.empty(shape, np.int8)
        stream = torch.cuda.current_stream(net.device).cuda_stream
        _lib.check(_lib.lib().tf2_net_read_layer(net._h, layer, batch, self._ws.data_ptr(), out.ctypes.data, out.size, stream))
        return out

    def Run(self):
        """runner.cpp:54-198: load the image file, run num_images frames, read back."""
        import torch
        net = self.network
        t = net.tables
        img = LoadInputImage(self.image_file, int(t["INPUT_IMAGE_C"]), int(t["INPUT_IMAGE_H"]), int(t["INPUT_IMAGE_W"]))
        batch = np.repeat(img[None], self.num_images, axis=0)
        x = torch.from_numpy(batch).to(net.device)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        logits = self.run_batch(x)
        end.record()
        torch.cuda.synchronize(net.device)
        ms = start.elapsed_time(end)
        self.latency_ms = ms
        self.throughput_fps = self.num_images / (ms * 1e-3) if ms > 0 else float("inf")
        net.output = logits.cpu().numpy()
        return net.output


def LoadInputImage(image_name, C=3, H=224, W=224) -> np.ndarray:
    """input_loader.cpp:76-96: raw float32 [C][H][W]."""
    if isinstance(image_name, np.ndarray):
        return np.ascontiguousarray(image_name, np.float32).reshape(C, H, W)
    a = np.fromfile(image_name, dtype=np.float32, count=C * H * W)
    if a.size != C * H * W:
        raise ValueError(f"load input image : {image_name} Error (need {C * H * W} floats, got {a.size})")
    return a.reshape(C, H, W)


def _last_q_row(q: np.ndarray, num_layer: Optional[int], n_out: int) -> np.ndarray:
    """The reference passes the WHOLE q table and indexes q[NUM_LAYER * MAX_OUT_CHANNEL + n]
    (main.cpp:52-53, network_helper.cpp:127,181): row NUM_LAYER = the Q of the last layer's
    output.  A 1-D array is taken as that row itself."""
    q = np.asarray(q)
    if q.ndim == 2:
        if num_layer is None:
            raise ValueError("a 2-D q table needs num_layer (the reference indexes row NUM_LAYER)")
        row = q[num_layer]
    elif q.ndim == 1:
        row = q
    else:
        raise ValueError("q must be the [NUM_Q_LAYERS][MAX_OUT_CHANNEL] table or its last-layer row")
    if row.size < n_out:
        raise ValueError(f"q row has {row.size} channels, the output {n_out}")
    return np.ascontiguousarray(row[:n_out], np.int8)


def Evaluation(n: int, q: np.ndarray, output: np.ndarray, k: int = 5, num_layer: Optional[int] = None):
    """network_helper.cpp:143-207: dequantise frame n's logits with the last layer's Q row
    (q = the full table + num_layer = NUM_LAYER, as main.cpp:53 passes it, or that row alone),
    top-k with the reference's tie rule, softmax probability.  Returns (labels, probabilities)."""
    logits = np.ascontiguousarray(output[n], np.int8).ravel()
    q_last = _last_q_row(q, num_layer, logits.size)
    labels = np.empty(k, np.int32)
    feats = np.empty(k, np.float32)
    _lib.check(_lib.lib().tf2_topk(logits.ctypes.data, q_last.ctypes.data, logits.size, k, labels.ctypes.data, feats.ctypes.data))
    trans = (1 << (-q_last.astype(np.int32))).astype(np.float32)
    f = logits.astype(np.float32) / trans
    sum_exp = np.float32(0)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        for v in f:                                # float accumulation order of the reference (:187)
            sum_exp = np.float32(sum_exp + np.exp(np.float32(v)))
        probs = np.exp(feats.astype(np.float32)) / sum_exp
    return labels.tolist(), probs.tolist()


def Verify(n: int, file_name, q: np.ndarray, output: np.ndarray, num_layer: Optional[int] = None) -> float:
    """network_helper.cpp:18-141: relative L1 error of frame n's int8 output against the golden
    float tensor scaled by 2^Q.  output[n] is [N] or NHWC [H*W][N]; the golden file is the
    reference's [N][H][W] float tensor; the per-CHANNEL factor 1 << -q[NUM_LAYER][c] (:127) is
    repeated over the H*W positions."""
    expect = np.fromfile(file_name, dtype=np.float32) if isinstance(file_name, (str, os.PathLike)) else np.asarray(file_name, np.float32)
    out = np.asarray(output[n])
    if out.ndim == 1:
        out = out[None, :]
    hw, N = out.shape[0] * int(np.prod(out.shape[1:-1], dtype=np.int64)), out.shape[-1]
    out = out.reshape(hw, N).astype(np.float32)
    q_last = _last_q_row(q, num_layer, N)
    trans = (1 << (-q_last.astype(np.int32))).astype(np.float32)            # per channel
    if expect.size < N * hw:
        raise ValueError(f"golden tensor has {expect.size} values, the output {N * hw}")
    et = expect.ravel()[:N * hw].reshape(N, hw).T * trans[None, :]         # [hw][N], channel-wise scale
    total_error = np.abs(et - out).astype(np.float32).sum(dtype=np.float32)
    total_expect = np.abs(et).astype(np.float32).sum(dtype=np.float32)
    return float(total_error / total_expect) if total_expect else float("inf")
