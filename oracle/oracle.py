"""TEST INFRASTRUCTURE -- ctypes/numpy front end of oracle/libtf2oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; tf2_amd/ (the product) never does.  See tf2_oracle.c for the
reference file:line each function follows.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libtf2oracle.so")
    src = os.path.join(_HERE, "tf2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return so


class LayerT(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "C", "H", "W", "N", "FH", "FW", "stride", "pad_h", "pad_w", "dil", "OH", "OW",
        "relu", "pool_en", "pool_S", "pool_st", "pool_pad", "PH", "PW",
        "add_en", "add_relu", "endpool", "endpool_mult")]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libtf2oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.tf2o_get_real.restype = C.c_uint8
        L.tf2o_get_real.argtypes = [C.c_float, C.c_int8]
        L.tf2o_mul.restype = C.c_int32
        L.tf2o_mul.argtypes = [C.c_int8, C.c_uint8]
        L.tf2o_num_threads.restype = C.c_int
        L.tf2o_set_num_threads.argtypes = [C.c_int]
        if "OMP_NUM_THREADS" not in os.environ:       # not more threads than CPUs this process may really use (cgroup quota)
            L.tf2o_set_num_threads(min(L.tf2o_num_threads(), effective_cpus()))
        _LIB = L
    return _LIB


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container that sees 256 logical
    CPUs with cpu.max = 16 cores runs 256 OpenMP threads on 16 cores' worth of time)."""
    import math, os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, math.ceil(q / p_)))
    except (OSError, ValueError):
        pass
    return n


def num_threads():
    """OpenMP threads the oracle runs on: the process's effective CPUs (set once), unless OMP_NUM_THREADS says otherwise."""
    return lib().tf2o_num_threads()


def get_real(w, expand):
    return int(lib().tf2o_get_real(float(np.float32(w)), int(expand)))


def mul(x, code):
    return int(lib().tf2o_mul(int(x), int(code)))


def encode_filters(w, q_in, q_out):
    """w float32 [N,C,FH,FW]; q_in int8[C], q_out int8[N] (runtime = negated Q)."""
    w = np.ascontiguousarray(w, np.float32)
    N, Cc, FH, FW = w.shape
    q_in = np.ascontiguousarray(q_in, np.int8); q_out = np.ascontiguousarray(q_out, np.int8)
    assert q_in.size >= Cc and q_out.size >= N
    codes = np.empty(w.shape, np.uint8)
    lib().tf2o_encode_filters(_p(w), N, Cc, FH, FW, _p(q_in), _p(q_out), _p(codes))
    return codes


def fold_bias_bn(N, q_out, bias=None, bn=None):
    """bn = (mean, var, scale_factor, gamma, beta) or None. -> bias_fix, alpha_fix, beta_fix int32[N]."""
    q_out = np.ascontiguousarray(q_out, np.int8)
    bf = np.empty(N, np.int32); af = np.empty(N, np.int32); btf = np.empty(N, np.int32)
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
    if bn is not None:
        mean, var, sf, gamma, beta = bn
        mean = np.ascontiguousarray(mean, np.float32); var = np.ascontiguousarray(var, np.float32)
        gamma = np.ascontiguousarray(gamma, np.float32); beta = np.ascontiguousarray(beta, np.float32)
    else:
        mean = var = gamma = beta = None; sf = 0.0
    lib().tf2o_fold_bias_bn(N, int(bias is not None), int(bn is not None), _p(bias), _p(mean), _p(var),
                            C.c_float(float(sf)), _p(gamma), _p(beta), _p(q_out), _p(bf), _p(af), _p(btf))
    return bf, af, btf


def conv1_rewrite(codes7):
    codes7 = np.ascontiguousarray(codes7, np.uint8)
    N = codes7.shape[0]
    assert codes7.shape[1:] == (3, 7, 7)
    out = np.empty((N, 27, 3, 3), np.uint8)
    lib().tf2o_conv1_rewrite(_p(codes7), N, _p(out))
    return out


def feature_trans(img):
    """img float32 [3,224,224] -> [27,114,114] (LoadInputImage, input_loader.cpp:76-118)."""
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty((27, 114, 114), np.float32)
    for c in range(3):
        plane = np.empty((9, 114, 114), np.float32)
        lib().tf2o_feature_trans(_p(img[c]), _p(plane))
        out[9 * c:9 * c + 9] = plane
    return out


def q_table(vals, tables, max_c=None, n_q_rows=None):
    """Q-file ints -> runtime q[n_q_rows, max_c] int8 (negated).  tables: dict of k* lists."""
    t = tables
    num_layer = int(t["NUM_LAYER"]); num_conv = int(t["NUM_CONVOLUTIONS"])
    max_c = int(max_c or t["MAX_OUT_CHANNEL"]); n_q_rows = int(n_q_rows or t["NUM_Q_LAYERS"])
    vals = np.ascontiguousarray(vals, np.int32)
    q = np.zeros((n_q_rows, max_c), np.int8)
    arr = lambda k: np.ascontiguousarray(t[k], np.int32)
    a = [arr(k) for k in ("kOutputChannels", "kIpoolEnable", "kInputLayer", "kBranchTail", "kConcatLayer", "kNStart")]
    used = lib().tf2o_q_table(_p(vals), vals.size, num_layer, num_conv, max_c, *[_p(x) for x in a], _p(q))
    return q, used


def quantize_input(x, q0):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.int8)
    lib().tf2o_quantize_input(_p(x), C.c_size_t(x.size), int(q0), _p(out))
    return out


def conv(x, codes, bias, stride=1, pad=0, dil=1):
    """x int8 [C,H,W]; codes uint8 [N,C,FH,FW]; bias int32[N] -> acc int32 [N,OH,OW]."""
    x = np.ascontiguousarray(x, np.int8); codes = np.ascontiguousarray(codes, np.uint8)
    bias = np.ascontiguousarray(bias, np.int32)
    Cc, H, W = x.shape; N, C2, FH, FW = codes.shape
    assert Cc == C2
    OH = (H + 2 * pad - dil * (FH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (FW - 1) - 1) // stride + 1
    acc = np.empty((N, OH, OW), np.int32)
    lib().tf2o_conv(_p(x), Cc, H, W, _p(codes), _p(bias), N, FH, FW, stride, pad, pad, dil, OH, OW, _p(acc))
    return acc


def requant(acc, alpha, beta, relu):
    acc = np.ascontiguousarray(acc, np.int32)
    N = acc.shape[0]; HW = acc.size // N
    alpha = np.ascontiguousarray(alpha, np.int32); beta = np.ascontiguousarray(beta, np.int32)
    y = np.empty(acc.shape, np.int8)
    lib().tf2o_requant(_p(acc), N, HW, _p(alpha), _p(beta), int(relu), _p(y))
    return y


def maxpool(x, S, st, pad, PH, PW):
    x = np.ascontiguousarray(x, np.int8)
    Cc, H, W = x.shape
    y = np.empty((Cc, PH, PW), np.int8)
    lib().tf2o_maxpool(_p(x), Cc, H, W, S, st, pad, PH, PW, _p(y))
    return y


def residual_add(y, res, relu):
    y = np.array(y, np.int8, copy=True, order="C"); res = np.ascontiguousarray(res, np.int8)
    lib().tf2o_residual_add(_p(y), _p(res), C.c_size_t(y.size), int(relu))
    return y


def global_avg(x, mult=669):
    x = np.ascontiguousarray(x, np.int8)
    Cc = x.shape[0]; HW = x.size // Cc
    y = np.empty(Cc, np.int8)
    lib().tf2o_global_avg(_p(x), Cc, HW, int(mult), _p(y))
    return y


def l2norm(x, qx, qy, w):
    """x int8 [C,H,W]; qx / qy runtime q rows (int8, negated file Q); w float32 [C] -> int8 [C,H,W]."""
    x = np.ascontiguousarray(x, np.int8)
    Cc = x.shape[0]; HW = x.size // Cc
    qx = np.ascontiguousarray(qx[:Cc], np.int8); qy = np.ascontiguousarray(qy[:Cc], np.int8)
    w = np.ascontiguousarray(w, np.float32)
    y = np.empty(x.shape, np.int8)
    lib().tf2o_l2norm(_p(x), Cc, HW, _p(qx), _p(qy), _p(w), _p(y))
    return y


def topk(logits, q, k=5):
    logits = np.ascontiguousarray(logits, np.int8); q = np.ascontiguousarray(q, np.int8)
    labels = np.empty(k, np.int32); feats = np.empty(k, np.float32)
    lib().tf2o_topk(_p(logits), _p(q), logits.size, k, _p(labels), _p(feats))
    return labels, feats


def layer(spec, x, codes, bias, alpha, beta, res=None):
    """Fused layer on a batch.  spec: dict with LayerT fields.  x int8 [B,C,H,W]."""
    L = LayerT(**{k: int(v) for k, v in spec.items()})
    x = np.ascontiguousarray(x, np.int8); B = x.shape[0]
    assert x.shape[1:] == (L.C, L.H, L.W), (x.shape, (L.C, L.H, L.W))
    codes = np.ascontiguousarray(codes, np.uint8)
    assert codes.shape == (L.N, L.C, L.FH, L.FW), (codes.shape, (L.N, L.C, L.FH, L.FW))
    bias = np.ascontiguousarray(bias, np.int32); alpha = np.ascontiguousarray(alpha, np.int32)
    beta = np.ascontiguousarray(beta, np.int32)
    if L.endpool:
        y = np.empty((B, L.N, 1, 1), np.int8)
    else:
        y = np.empty((B, L.N, L.PH, L.PW), np.int8)
    if res is not None:
        res = np.ascontiguousarray(res, np.int8)
        assert res.shape == (B, L.N, L.PH, L.PW)
    lib().tf2o_layer(C.byref(L), B, _p(x), _p(codes), _p(bias), _p(alpha), _p(beta), _p(res), _p(y))
    return y
