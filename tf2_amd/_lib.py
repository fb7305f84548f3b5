"""ctypes binding of the C-ABI library (include/tf2_amd.h).

The product path FAILS LOUDLY when the HIP extension is missing: there is no CPU
fallback anywhere in tf2_amd (the CPU restatement lives in oracle/ and is test
infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import sys
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# One hardware queue per in-flight stream: the HIP runtime multiplexes streams onto 4 hardware queues by default (one of them the
# null stream's), so a fourth in-flight batch would queue behind another (-15 %, profiles/r02_inflight_hwqueues.txt).  The runtime
# reads the variable when it initialises, i.e. it only takes effect if tf2_amd is imported before the first HIP call of the
# process; a deployment that embeds the C library directly exports it itself (INTEGRATION.md "Deployment preconditions").
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    _torch_mod = sys.modules.get("torch")
    if _torch_mod is not None and getattr(getattr(_torch_mod, "cuda", None), "is_initialized", lambda: False)():
        import warnings
        warnings.warn("tf2_amd: HIP was initialised before tf2_amd was imported, GPU_MAX_HW_QUEUES=8 cannot take effect any more: a fourth "
                      "batch in flight will share a hardware queue (export GPU_MAX_HW_QUEUES=8 before starting the process)", RuntimeWarning)
# TF2_AMD_LIB: another build of the same sources (the tools' -DTF2_PROBES / -DTF2_CHECK_DMA libraries); never a fallback, and refused
# unless the process also sets TF2_AMD_TOOL_LIB=1 (lib() below: a tool build can leave work out of a step)
LIB_PATH = os.environ.get("TF2_AMD_LIB") or os.path.join(_HERE, "libtf2amd.so")


class Tf2Error(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"tf2_amd status {status}: {msg}")
        self.status = status


class LayerDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "src", "q_in_row", "C", "H", "W", "N", "k", "stride", "pad_h", "pad_w", "dil", "OH", "OW",
        "bias_en", "bn_en", "relu", "ipool", "pool_en", "pool_S", "pool_st", "pool_pad", "PH", "PW",
        "add_src", "add_relu", "endpool", "endpool_mult", "concat", "n_start", "model_C", "model_k")]


class RunOpts(C.Structure):
    """tf2_run_opts (include/tf2_amd.h)."""
    _fields_ = [("size", C.c_uint32), ("images_are_q", C.c_int32), ("concurrency", C.c_int32), ("mark_after_layer", C.c_int32),
                ("mark_event", C.c_void_p)]


class LaunchInfo(C.Structure):
    """tf2_launch_info (include/tf2_amd.h)."""
    _fields_ = [("layer", C.c_int32), ("grid", C.c_int32), ("block", C.c_int32), ("lds_bytes", C.c_int32), ("vgprs", C.c_int32),
                ("kernel", C.c_char * 96)]


class TensorInfo(C.Structure):
    """tf2_tensor_info (include/tf2_amd.h)."""
    _fields_ = [("offset", C.c_int64), ("bytes", C.c_int64), ("first_row", C.c_int32), ("last_row", C.c_int32)]


class RowTensors(C.Structure):
    """tf2_row_tensors (include/tf2_amd.h)."""
    _fields_ = [("in_tensor", C.c_int32), ("out_tensor", C.c_int32), ("conv_tensor", C.c_int32), ("res_tensor", C.c_int32)]


class NetDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_layers", "n_conv", "n_q_rows", "max_out_channel", "image_c", "image_h", "image_w",
        "conv1_rewrite", "n_concat")]


def build(force: bool = False) -> str:
    """Compile tf2_amd/libtf2amd.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".cpp", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "tf2_amd.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", src_dir])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension was not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C tf2_amd/csrc`). tf2_amd has no CPU fallback by design.")
    # The HIP runtime of the process must be ONE: PyTorch-ROCm ships its own libamdhip64, and libtf2amd.so loaded BEFORE torch binds the
    # system copy under /opt/rocm instead -- a later torch import then brings the second runtime, and the library's first launch reports
    # "no ROCm-capable device is detected" (seen with build() + smoke() in one process, round 6).  Loading torch first makes the
    # dynamic loader resolve libtf2amd.so's HIP symbols against the runtime torch loaded.  (Hosts without torch -- the C++ shim of
    # INTEGRATION.md section 3 -- link one runtime anyway.)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.tf2_build_kind.restype = C.c_int
    kind = L.tf2_build_kind()
    if kind != 0 and os.environ.get("TF2_AMD_TOOL_LIB") != "1":
        raise ImportError(f"{LIB_PATH} is a TOOL build of the library (tf2_build_kind() = {kind}: 1 timing probes, 2 DMA check), not the "
                          "product; tools that want it set TF2_AMD_TOOL_LIB=1 next to TF2_AMD_LIB")
    vp, sz, i32p, i8p, u8p, fp = C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int8), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    L.tf2_last_error.restype = C.c_char_p
    L.tf2_abi_version.restype = C.c_int
    L.tf2_has_device_code.restype = C.c_int
    L.tf2_get_real.restype = C.c_uint8
    L.tf2_get_real.argtypes = [C.c_float, C.c_int8]
    L.tf2_net_create.argtypes = [C.POINTER(NetDesc), C.POINTER(LayerDesc), C.POINTER(vp)]
    L.tf2_net_destroy.argtypes = [vp]
    L.tf2_net_destroy.restype = None
    L.tf2_quantization.argtypes = [vp, C.c_char_p, sz, vp, sz, i32p]
    L.tf2_net_set_q.argtypes = [vp, vp, sz]
    L.tf2_net_load_model.argtypes = [vp, vp, sz]
    L.tf2_model4bit_decode.argtypes = [vp, sz, vp, sz, vp]
    L.tf2_net_load_model_4bit.argtypes = [vp, vp, sz]
    L.tf2_net_get_codes.argtypes = [vp, C.c_int, vp, sz, C.POINTER(sz)]
    L.tf2_net_get_bias_bn.argtypes = [vp, C.c_int, vp, vp, vp, sz]
    L.tf2_net_pack.argtypes = [vp, C.c_int]
    L.tf2_net_packed_size.argtypes = [vp]
    L.tf2_net_packed_size.restype = sz
    L.tf2_net_packed_copy.argtypes = [vp, vp, sz]
    L.tf2_net_packed_adopt.argtypes = [vp, vp, sz]
    L.tf2_net_bind_device.argtypes = [vp, vp, sz]
    L.tf2_net_workspace_size.argtypes = [vp, C.c_int, C.c_int]
    L.tf2_net_workspace_size.restype = sz
    L.tf2_net_logits_size.argtypes = [vp, C.c_int]
    L.tf2_net_logits_size.restype = sz
    L.tf2_net_reload_options.argtypes = [vp]
    L.tf2_net_run.argtypes = [vp, vp, C.c_int, vp, sz, vp, vp]
    L.tf2_net_run_q.argtypes = [vp, vp, C.c_int, vp, sz, vp, vp]
    L.tf2_net_run_ex.argtypes = [vp, vp, C.c_int, vp, sz, vp, vp, C.POINTER(RunOpts)]
    L.tf2_net_run_stats.argtypes = [vp, vp]
    L.tf2_net_poll_error.argtypes = [vp, C.c_int, vp, C.c_size_t, vp]
    L.tf2_net_describe_launches.argtypes = [vp, C.c_int, C.c_int, C.POINTER(LaunchInfo), C.c_int, C.POINTER(C.c_int)]
    L.tf2_net_describe_workspace.argtypes = [vp, C.c_int, C.c_int, C.POINTER(TensorInfo), C.c_int, C.POINTER(C.c_int), C.POINTER(RowTensors), C.c_int]
    L.tf2_net_read_layer.argtypes = [vp, C.c_int, C.c_int, vp, vp, sz, vp]
    L.tf2_net_profile.argtypes = [vp, C.c_int]
    L.tf2_net_profile_read.argtypes = [vp, vp, vp, vp, C.c_int]
    L.tf2_net_profile_loop_read.argtypes = [vp, vp, vp]
    L.tf2_topk.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    _lib = L
    return L


EXPORTED = [
    "tf2_last_error", "tf2_abi_version", "tf2_has_device_code", "tf2_build_kind", "tf2_get_real", "tf2_quantization",
    "tf2_net_create", "tf2_net_destroy", "tf2_net_set_q", "tf2_net_load_model", "tf2_model4bit_decode", "tf2_net_load_model_4bit", "tf2_net_get_codes",
    "tf2_net_get_bias_bn", "tf2_net_pack", "tf2_net_packed_size", "tf2_net_packed_copy",
    "tf2_net_packed_adopt", "tf2_net_bind_device", "tf2_net_workspace_size", "tf2_net_logits_size", "tf2_net_reload_options", "tf2_net_run",
    "tf2_net_run_q", "tf2_net_run_ex", "tf2_net_run_stats", "tf2_net_poll_error", "tf2_net_describe_launches", "tf2_net_describe_workspace", "tf2_net_read_layer", "tf2_net_profile", "tf2_net_profile_read", "tf2_net_profile_loop_read", "tf2_topk"]


def parse_opts(text: str) -> dict:
    """TF2_AMD_OPTS text -> {name: value} exactly as csrc/opts.cpp reads it: items separated by ',', ';' or ' ', a bare name means
    name=1.  (An empty value, 'name=', is an error on the C side; it is kept here so that the library reports it.)"""
    cur = {}
    for item in re.split(r"[,; ]", text or ""):
        if not item:
            continue
        name, eq, val = item.partition("=")
        cur[name] = val if eq else "1"
    return cur


def set_opts(**kw) -> None:
    """Tools: set / change / remove (value None) options of TF2_AMD_OPTS (csrc/opts.h) in this process's environment, cumulatively, and
    admit the test-only ones (TF2_AMD_TEST=1).  Takes effect at the next tf2_net_create / NetWork.reload_options()."""
    cur = parse_opts(os.environ.get("TF2_AMD_OPTS", ""))
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    os.environ["TF2_AMD_TEST"] = "1"
    os.environ["TF2_AMD_OPTS"] = ",".join(f"{k}={v}" for k, v in cur.items())


def check(status: int) -> None:
    if status != 0:
        raise Tf2Error(status, lib().tf2_last_error().decode("utf-8", "replace"))
