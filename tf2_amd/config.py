"""Network-config ingestion for the Runtime_Engine/cnn drop-in.

The reference compiles one generated C header per network (``<net>.h``, written by
``TF2_auto_config`` from ``fpganetwork.bin``) into both its host and its OpenCL device
code (reference: Runtime_Engine/cnn/host/inc/cnn.h:29-35, resnet50.h:119-1372).  The
k* tables of that header ARE the layer program.  This module keeps that format as the
interface:

* :func:`parse_net_header`   -- macro-aware reader of a ``<net>.h`` (the unchanged
  TF2_auto_config output, or the hand-tuned shipped headers) -> :class:`NetTables`.
* :func:`read_fpganetwork` / :func:`tables_from_fpganetwork` -- direct reader of the
  ``fpganetwork.bin`` struct dump (writer: TransForm_Kit/ModelConvert/caffe2fpga/src/
  tools.cpp:437-568; reader: Runtime_Engine/TF2_auto_config/src/fpganetworkinterface.cpp:6-169)
  producing the same table set without the C++ tool.
* :func:`resnet50_tables`, :func:`squeezenet11_tables`, :func:`vgg16_tables`,
  :func:`tiny_tables` -- programmatic builders in the same vocabulary (the reference
  ships headers for ResNet50/GoogLeNet only).
* :func:`build_plan` -- turns tables into an explicit layer plan (input tensor,
  residual source, concat slice, conv/pool geometry) that the executor walks.

Only the semantic tables are used; the FPGA schedule (cache pages, cycle tables,
vector re-layouts) is not reproduced.
"""
from __future__ import annotations

import re
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

# Architecture constants of the reference (host/inc/archs.h:25-43, types.h:31-35).
ARCH = dict(IMAGE_BATCH_SIZE=1, N_VECTOR=16, C_VECTOR=16, OW_VECTOR=5, FW_VECTOR=3,
            NARROW_N_VECTOR=16, W_VECTOR=7, DOUBLE_BUFFER_DIM=2, DDR_BANDWIDTH_IN_BYTES=64,
            REALMAX=127, REALMIN=-128, ALPHA_INFLAT=20, INFLAT=15, NUM_IMAGES=1)
INFLAT = 15
ALPHA_INFLAT = 20

# Per-layer tables the engine consumes (all others in a header are parsed but ignored).
LAYER_KEYS = (
    "kFilterSize", "kPadWidth", "kPadHeight", "kInputWidth", "kInputHeight", "kOutputWidth",
    "kOutputHeight", "kInputChannels", "kOutputChannels", "kConvStride", "kIpoolEnable",
    "kPoolEnable", "kBiasEnable", "kBnEnable", "kReluEnable", "kPoolWindow", "kPoolStride2",
    "kPoolOutputWidth", "kPoolOutputHeight", "kPoolPad", "kAdditionEnable", "kAdditionReluEnable",
    "kEndPoolEnable", "kInputLayer", "kNStart", "kNEnd", "kBranchTail", "kConcatLayer",
    "kDDRReadBase", "kDDRWriteBase", "kDDRWriteEnable")
SCALAR_KEYS = ("NUM_LAYER", "NUM_CONVOLUTIONS", "NUM_Q_LAYERS", "INPUT_IMAGE_C", "INPUT_IMAGE_H",
               "INPUT_IMAGE_W", "FIRST_FILTER_SIZE", "MAX_OUT_CHANNEL")


class ConfigError(ValueError):
    pass


class NetTables(dict):
    """dict of scalars (NUM_LAYER, ...) and per-layer int lists (kFilterSize, ...).

    Extension keys (absent from reference headers, default 0/1) let the builders
    describe networks the reference never shipped a header for:
      xConv1Rewrite   1 -> layer 0 is the 7x7/s2 conv executed as 27-ch 3x3 on the
                      114x114 space-to-depth image (model_loader.cpp:244-257,
                      input_loader.cpp:98-116); the reference does this for
                      ResNet50 and GoogLeNet unconditionally.
      xDilation[l]    conv dilation (SSD conv6), default 1.
      xEndPoolMult[l] fixed-point reciprocal of the global-average window,
                      default 669 = round(2^15/49) (full_size_pool.cl:115-118).
      xResidualSrc[l] explicit residual producer layer (else derived from the DDR
                      page plan), -1 for none.
    """

    @property
    def num_layer(self) -> int:
        return int(self["NUM_LAYER"])

    def validate(self) -> "NetTables":
        n = self.num_layer
        for k in LAYER_KEYS:
            if k not in self:
                raise ConfigError(f"network tables lack {k}")
            if len(self[k]) != n:
                raise ConfigError(f"{k} has {len(self[k])} entries, NUM_LAYER is {n}")
        for k in SCALAR_KEYS:
            if k not in self:
                raise ConfigError(f"network tables lack {k}")
        return self


# ----------------------------------------------------------------------------------
# A small C constant-expression evaluator (enough for the generated headers:
# integers, identifiers, + - * / % << >> comparisons && || ! ?: and the macros of
# host/inc/defines.h:47-55 -- CEIL, NEXT_DIVISIBLE, NEXT_POWER_OF_2, MYMAX2).
# ----------------------------------------------------------------------------------
_TOK = re.compile(r"\s*(?:(\d+)[uUlL]*|([A-Za-z_]\w*)|(<<|>>|<=|>=|==|!=|&&|\|\||[-+*/%()<>?:!,&|^~]))")


def _next_pow2(x: int) -> int:
    x -= 1
    for s in (1, 2, 4, 8, 16):
        x |= x >> s
    return x + 1


_BUILTIN_FUNCS = {
    "CEIL": lambda x, y: (x - 1) // y + 1,
    "MYCEIL": lambda x, y: (x - 1) // y + 1,
    "NEXT_DIVISIBLE": lambda x, y: x if x % y == 0 else x + y - x % y,
    "NEXT_POWER_OF_2": _next_pow2,
    "MYMAX2": lambda x, y: x if x >= y else y,
}


class _Expr:
    def __init__(self, text: str, env: Dict[str, object]):
        self.toks: List[str] = []
        pos = 0
        text = text.strip()
        while pos < len(text):
            m = _TOK.match(text, pos)
            if not m:
                raise ConfigError(f"cannot tokenise expression {text!r} at {pos}")
            self.toks.append(m.group(1) or m.group(2) or m.group(3))
            pos = m.end()
        self.i = 0
        self.env = env

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else None

    def eat(self, t=None):
        tok = self.peek()
        if t is not None and tok != t:
            raise ConfigError(f"expected {t!r}, got {tok!r} in {' '.join(self.toks)}")
        self.i += 1
        return tok

    def parse(self) -> int:
        v = self.ternary()
        if self.peek() is not None:
            raise ConfigError(f"trailing tokens in expression: {' '.join(self.toks)}")
        return v

    def ternary(self) -> int:
        c = self.binary(0)
        if self.peek() == "?":
            self.eat()
            a = self.ternary()
            self.eat(":")
            b = self.ternary()
            return a if c else b
        return c

    _PREC = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="),
             ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def binary(self, lvl: int) -> int:
        if lvl == len(self._PREC):
            return self.unary()
        v = self.binary(lvl + 1)
        while self.peek() in self._PREC[lvl]:
            op = self.eat()
            r = self.binary(lvl + 1)
            v = self._apply(op, v, r)
        return v

    @staticmethod
    def _apply(op, a, b):
        if op == "+": return a + b
        if op == "-": return a - b
        if op == "*": return a * b
        if op == "/":
            if b == 0: raise ConfigError("division by zero in header expression")
            q = abs(a) // abs(b)
            return q if (a >= 0) == (b >= 0) else -q
        if op == "%":
            return a - b * _Expr._apply("/", a, b)
        if op == "<<": return a << b
        if op == ">>": return a >> b
        if op == "<": return int(a < b)
        if op == ">": return int(a > b)
        if op == "<=": return int(a <= b)
        if op == ">=": return int(a >= b)
        if op == "==": return int(a == b)
        if op == "!=": return int(a != b)
        if op == "&&": return int(bool(a) and bool(b))
        if op == "||": return int(bool(a) or bool(b))
        if op == "&": return a & b
        if op == "|": return a | b
        if op == "^": return a ^ b
        raise ConfigError(op)

    def unary(self) -> int:
        t = self.peek()
        if t == "-":
            self.eat(); return -self.unary()
        if t == "+":
            self.eat(); return self.unary()
        if t == "!":
            self.eat(); return int(not self.unary())
        if t == "~":
            self.eat(); return ~self.unary()
        return self.primary()

    def primary(self) -> int:
        t = self.eat()
        if t is None:
            raise ConfigError("unexpected end of expression")
        if t == "(":
            v = self.ternary()
            self.eat(")")
            return v
        if t.isdigit():
            return int(t)
        if t in ("true", "false"):
            return int(t == "true")
        if t in _BUILTIN_FUNCS and self.peek() == "(":
            self.eat("(")
            args = [self.ternary()]
            while self.peek() == ",":
                self.eat(); args.append(self.ternary())
            self.eat(")")
            return _BUILTIN_FUNCS[t](*args)
        if t in self.env:
            v = self.env[t]
            if isinstance(v, str):            # object-like macro: evaluate lazily
                v = _Expr(v, self.env).parse()
                self.env[t] = v
            if isinstance(v, list):
                raise ConfigError(f"array {t} used as scalar")
            return int(v)
        raise ConfigError(f"unknown identifier {t!r} in header expression")


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def parse_net_header(text_or_path: str, defines: Optional[Sequence[str]] = None) -> NetTables:
    """Parse a TF2 network header (``resnet50.h`` format) into :class:`NetTables`.

    Understands ``#define`` object-like macros (lazily evaluated), ``#ifdef/#ifndef/
    #else/#endif`` on plain names, and ``CONSTANT <type> name[...] = {...};`` /
    ``CONSTANT <type> name = expr;`` definitions (resnet50.h:46-113, :119-1372).
    Function-like macros and anything after the tables that cannot be evaluated
    (cycle helpers) are skipped.
    """
    text = text_or_path
    if "\n" not in text_or_path and len(text_or_path) < 4096:
        with open(text_or_path, "r", errors="replace") as f:
            text = f.read()
    text = _strip_comments(text).replace("\\\n", " ")
    env: Dict[str, object] = dict(ARCH)
    defined = set(defines or ())
    out = NetTables()
    # -- pass 1: preprocessor (line based) --------------------------------------
    kept: List[str] = []
    stack: List[bool] = []
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            d = s[1:].strip()
            m = re.match(r"(ifdef|ifndef)\s+(\w+)", d)
            if m:
                have = m.group(2) in defined or m.group(2) in env
                stack.append(have if m.group(1) == "ifdef" else not have)
                continue
            if re.match(r"if\b", d):
                stack.append(False)           # '#if expr' blocks are debug-only in these headers
                continue
            if d.startswith("else"):
                if stack: stack[-1] = not stack[-1]
                continue
            if d.startswith("elif"):
                if stack: stack[-1] = False
                continue
            if d.startswith("endif"):
                if stack: stack.pop()
                continue
            if not all(stack):
                continue
            m = re.match(r"define\s+(\w+)(\()?\s*(.*)$", d)
            if m:
                name, fparen, body = m.group(1), m.group(2), m.group(3).strip()
                if fparen is not None:
                    continue                  # function-like macro (cycle helpers): not a table
                defined.add(name)
                if body:
                    env[name] = body
                else:
                    env.setdefault(name, 1)
            continue
        if all(stack):
            kept.append(line)
    body = "\n".join(kept)
    # -- pass 2: CONSTANT definitions ----------------------------------------------
    for m in re.finditer(r"\b(?:CONSTANT|static\s+const|constant)\s+(?:unsigned\s+)?(\w+)\s+(\w+)\s*"
                         r"(\[[^\]]*\])?\s*=\s*(\{.*?\}|[^;{]+);", body, flags=re.S):
        name, is_arr, rhs = m.group(2), m.group(3), m.group(4).strip()
        try:
            if is_arr or rhs.startswith("{"):
                inner = rhs.strip()[1:-1]
                items = [x.strip() for x in _split_top(inner) if x.strip()]
                vals = [_Expr(x, env).parse() for x in items]
                dim = (is_arr or "[]")[1:-1].strip()
                if dim:                        # C zero-fills a short initialiser list
                    want = _Expr(dim, env).parse()
                    if len(vals) < want:
                        vals += [0] * (want - len(vals))
                env[name] = vals
                out[name] = vals
            else:
                v = _Expr(rhs, env).parse()
                env[name] = v
                out[name] = v
        except ConfigError:
            continue                           # e.g. struct initialisers, cycle helpers
    for k in list(env):
        if isinstance(env[k], str):
            try:
                env[k] = _Expr(env[k], env).parse()
            except ConfigError:
                continue
    for k, v in env.items():
        if isinstance(v, int) and k not in ARCH and k not in out:
            out[k] = v
    out.setdefault("xConv1Rewrite", 1 if out.get("FIRST_FILTER_SIZE") == 7 else 0)
    return out.validate()


def _split_top(s: str) -> List[str]:
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return parts


# ----------------------------------------------------------------------------------
# fpganetwork.bin (raw LP64 struct dump)
# ----------------------------------------------------------------------------------
OP_CONV, OP_FC, OP_BN, OP_SCALE, OP_LRN, OP_RELU = 1, 2, 3, 4, 5, 6
OP_POOL, OP_ELTWISE, OP_CONCAT, OP_FLATTEN, OP_SOFTMAX = 11, 12, 13, 14, 15
FPGANET_VERSION = 20190510


@dataclass
class FpgaOp:
    type: int
    p: dict = field(default_factory=dict)


@dataclass
class FpgaLayer:
    id: int
    in_ids: List[int]
    out_ids: List[int]
    in_dims: List[tuple]
    out_dims: List[tuple]
    ops: List[FpgaOp]


def read_fpganetwork(path_or_bytes) -> List[FpgaLayer]:
    """Decode ``fpganetwork.bin`` (fpganetworkinterface.h:27-247 struct sizes on LP64:
    StFpgaNetInfo 144 B, StFpgaLayerInfo 424 B, StFpgaOpInfo 16 B, Conv 48, Fc 8, Bn 4,
    Scale 1, Pool 40, Eltwise 4)."""
    d = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if len(d) < 144:
        raise ConfigError("fpganetwork.bin: file too short")
    ver, = struct.unpack_from("<i", d, 0)
    nl, = struct.unpack_from("<i", d, 132)
    if ver != FPGANET_VERSION:
        raise ConfigError(f"fpganetwork.bin: version {ver}, expected {FPGANET_VERSION}")
    off = 144
    layers: List[FpgaLayer] = []
    for _ in range(nl):
        if off + 424 > len(d):
            raise ConfigError("fpganetwork.bin: truncated layer record")
        lid, nop, nin, nout = struct.unpack_from("<4i", d, off)
        in_ids = list(struct.unpack_from("<10i", d, off + 16))[:nin]
        out_ids = list(struct.unpack_from("<10i", d, off + 56))[:nout]
        in_dims = [struct.unpack_from("<4i", d, off + 96 + 16 * k) for k in range(nin)]
        out_dims = [struct.unpack_from("<4i", d, off + 256 + 16 * k) for k in range(nout)]
        off += 424
        ops = []
        for _o in range(nop):
            t, = struct.unpack_from("<i", d, off)
            off += 16
            if t == OP_CONV:
                v = struct.unpack_from("<i?3x4if5i", d, off); off += 48
                ops.append(FpgaOp(t, dict(out=v[0], bias=bool(v[1]), pad_l=v[2], pad_t=v[3], pad_r=v[4],
                                          pad_b=v[5], kh=v[7], kw=v[8], sh=v[9], sw=v[10], dil=v[11])))
            elif t == OP_FC:
                v = struct.unpack_from("<i?", d, off); off += 8
                ops.append(FpgaOp(t, dict(out=v[0], bias=bool(v[1]))))
            elif t == OP_BN:
                v = struct.unpack_from("<f", d, off); off += 4
                ops.append(FpgaOp(t, dict(eps=v[0])))
            elif t == OP_SCALE:
                v = struct.unpack_from("<?", d, off); off += 1
                ops.append(FpgaOp(t, dict(bias=bool(v[0]))))
            elif t == OP_POOL:
                v = struct.unpack_from("<9i?", d, off); off += 40
                ops.append(FpgaOp(t, dict(method=v[0], pad_l=v[1], pad_t=v[2], pad_r=v[3], pad_b=v[4],
                                          kh=v[5], kw=v[6], sh=v[7], sw=v[8], global_pool=bool(v[9]))))
            elif t == OP_ELTWISE:
                v = struct.unpack_from("<i", d, off); off += 4
                ops.append(FpgaOp(t, dict(method=v[0])))
            else:
                ops.append(FpgaOp(t))
        layers.append(FpgaLayer(lid, in_ids, out_ids, in_dims, out_dims, ops))
    return layers


def _blank_tables(n: int) -> NetTables:
    t = NetTables()
    for k in LAYER_KEYS:
        t[k] = [0] * n
    t["xDilation"] = [1] * n
    t["xEndPoolMult"] = [669] * n
    t["xResidualSrc"] = [-1] * n
    t["NUM_LAYER"] = t["NUM_CONVOLUTIONS"] = n
    return t


def tables_from_fpganetwork(layers: Sequence[FpgaLayer], netname: str = "resnet50") -> NetTables:
    """Fuse the op graph into conv blocks the way TF2_auto_config does
    (tf2_auto_param.cpp:152-893: conv+bn+scale+relu, a following max-pool folded into
    its producer, eltwise(+relu) folded into the LAST-computed addend, global average
    folded as kEndPoolEnable, fc as a 1x1 conv) and emit the k* tables.

    ``netname`` "resnet50"/"googlenet" selects the reference's conv1 convention
    (27x114x114 input, 3x3/s1/p0 filter, pool pad 1 -- tf2_auto_param.cpp:214-254,1593).
    Only straight-line + residual graphs are handled here (concat graphs -> use a
    generated header).
    """
    blocks: List[dict] = []          # one per conv/fc layer, in file order
    blob_owner: Dict[int, int] = {-1: -1}   # blob id -> producing conv block (-1 = image)
    for L in layers:
        kinds = [o.type for o in L.ops]
        if OP_CONV in kinds or OP_FC in kinds:
            op = L.ops[0]
            cin = L.in_dims[0]
            cout = L.out_dims[0]
            b = dict(src=blob_owner[L.in_ids[0]], C=cin[1], H=cin[2], W=cin[3], N=cout[1],
                     k=op.p.get("kh", 1), stride=op.p.get("sh", 1), pad=op.p.get("pad_l", 0),
                     dil=max(1, op.p.get("dil", 1)), bias=int(op.p.get("bias", False)),
                     bn=int(OP_BN in kinds), relu=int(OP_RELU in kinds), OH=cout[2], OW=cout[3],
                     pool=None, add=None, add_relu=0, endpool=0, is_fc=int(OP_FC in kinds))
            if b["is_fc"]:
                b.update(k=1, stride=1, pad=0, H=1, W=1, OH=1, OW=1, C=cin[1] * cin[2] * cin[3])
            blocks.append(b)
            blob_owner[L.out_ids[0]] = len(blocks) - 1
        elif kinds and kinds[0] == OP_POOL:
            p = L.ops[0].p
            owner = blob_owner[L.in_ids[0]]
            if owner < 0:
                raise ConfigError("pool on the raw image is not supported")
            if p["method"] == 1 or p["global_pool"]:     # average -> end pool
                blocks[owner]["endpool"] = 1
                blocks[owner]["endpool_hw"] = L.in_dims[0][2] * L.in_dims[0][3]
            else:
                blocks[owner]["pool"] = dict(S=p["kh"], stride=p["sh"], pad=p["pad_l"],
                                             PH=L.out_dims[0][2], PW=L.out_dims[0][3])
            blob_owner[L.out_ids[0]] = owner
        elif kinds and kinds[0] == OP_ELTWISE:
            a, bb = blob_owner[L.in_ids[0]], blob_owner[L.in_ids[1]]
            last, other = (a, bb) if a > bb else (bb, a)
            blocks[last]["add"] = other
            blocks[last]["add_relu"] = int(OP_RELU in kinds)
            blob_owner[L.out_ids[0]] = last
        elif kinds and kinds[0] in (OP_SOFTMAX, OP_RELU, OP_FLATTEN):
            blob_owner[L.out_ids[0]] = blob_owner[L.in_ids[0]]
        elif not kinds:
            blob_owner[L.out_ids[0]] = blob_owner[L.in_ids[0]]
        else:
            raise ConfigError(f"unsupported op sequence {kinds} in fpganetwork layer {L.id}")
    n = len(blocks)
    t = _blank_tables(n)
    rewrite = netname in ("resnet50", "googlenet") and blocks[0]["k"] == 7
    first = blocks[0]
    t.update(INPUT_IMAGE_C=first["C"], INPUT_IMAGE_H=first["H"], INPUT_IMAGE_W=first["W"],
             FIRST_FILTER_SIZE=first["k"], NUM_Q_LAYERS=n + 1, xConv1Rewrite=int(rewrite),
             MAX_OUT_CHANNEL=max(b["N"] for b in blocks))
    for i, b in enumerate(blocks):
        k, s, pad, H, W, C = b["k"], b["stride"], b["pad"], b["H"], b["W"], b["C"]
        if i == 0 and rewrite:
            # 7x7/s2/p3 over 3x224x224  ==  3x3/s1/p0 over 27x114x114
            C, H, W, k, s, pad = 27, 114, 114, 3, 1, 0
        oh1 = (H + 2 * pad - b["dil"] * (k - 1) - 1) + 1       # stride-1 output size (table convention)
        ow1 = (W + 2 * pad - b["dil"] * (k - 1) - 1) + 1
        t["kFilterSize"][i] = k; t["kPadWidth"][i] = t["kPadHeight"][i] = pad
        t["kInputWidth"][i] = W; t["kInputHeight"][i] = H
        t["kOutputWidth"][i] = ow1; t["kOutputHeight"][i] = oh1
        t["kInputChannels"][i] = C; t["kOutputChannels"][i] = b["N"]; t["kConvStride"][i] = s
        t["kBiasEnable"][i] = b["bias"]; t["kBnEnable"][i] = b["bn"]; t["kReluEnable"][i] = b["relu"]
        t["xDilation"][i] = b["dil"]
        t["kInputLayer"][i] = b["src"] + 1
        t["kNEnd"][i] = b["N"]
        oh, ow = (oh1 - 1) // s + 1, (ow1 - 1) // s + 1
        t["kPoolWindow"][i] = 3
        if b["pool"]:
            p = b["pool"]
            t["kPoolEnable"][i] = 1; t["kPoolWindow"][i] = p["S"]; t["kPoolStride2"][i] = int(p["stride"] == 2)
            ppad = p["pad"]
            if i == 0 and netname == "resnet50":
                ppad = 1                                    # tf2_auto_param.cpp:1593
            t["kPoolPad"][i] = ppad
            t["kPoolOutputHeight"][i] = p["PH"]; t["kPoolOutputWidth"][i] = p["PW"]
        else:
            t["kPoolOutputHeight"][i] = oh; t["kPoolOutputWidth"][i] = ow
        if b["add"] is not None:
            t["kAdditionEnable"][i] = 1; t["kAdditionReluEnable"][i] = b["add_relu"]
            t["xResidualSrc"][i] = b["add"]
        if b["endpool"]:
            t["kEndPoolEnable"][i] = 1
            hw = b.get("endpool_hw", 49)
            t["xEndPoolMult"][i] = 669 if hw == 49 else int(round(32768.0 / hw))
    _assign_ddr_pages(t)
    return t.validate()


def _assign_ddr_pages(t: NetTables) -> None:
    """Reproduce the reference's two-page DDR ping-pong for residual tensors
    (resnet50.h:170-268): every tensor that some later layer adds is written to DDR,
    alternating pages; the adder reads the page its addend was written to."""
    n = t.num_layer
    page = max(1, -(-256 // 16) * 56 * -(-56 // 7))     # DDR_PAGE_SIZE0 of resnet50.h:62 (7168)
    needed = {s for s in t["xResidualSrc"] if s >= 0}
    where: Dict[int, int] = {}
    for i in range(n):
        src = t["xResidualSrc"][i]
        if src >= 0:
            t["kDDRReadBase"][i] = where[src]
        if i in needed:
            base = (page - t["kDDRReadBase"][i]) if src >= 0 else 0
            t["kDDRWriteBase"][i] = base
            t["kDDRWriteEnable"][i] = 1
            where[i] = base


# ----------------------------------------------------------------------------------
# Explicit layer plan
# ----------------------------------------------------------------------------------
@dataclass
class LayerSpec:
    index: int
    # input
    src: int                 # producing layer (-1 = image, >=0 = layer, <=-2 = concat tensor -(k+2))
    q_in_row: int            # row of the Q table describing the input channels
    C: int; H: int; W: int
    # conv
    N: int; k: int; stride: int; pad_h: int; pad_w: int; dil: int
    OH: int; OW: int
    bias_en: int; bn_en: int; relu: int
    ipool: int
    # post ops
    pool_en: int; pool_S: int; pool_st: int; pool_pad: int
    PH: int; PW: int
    add_src: int; add_relu: int
    endpool: int; endpool_mult: int
    # output placement
    concat: int              # -1 or concat tensor id
    n_start: int
    model_C: int             # filter dims as stored in the model file (layer 0: image C / first filter)
    model_k: int

    def oracle_spec(self) -> dict:
        return dict(C=self.C, H=self.H, W=self.W, N=self.N, FH=self.k, FW=self.k, stride=self.stride,
                    pad_h=self.pad_h, pad_w=self.pad_w, dil=self.dil, OH=self.OH, OW=self.OW,
                    relu=self.relu, pool_en=self.pool_en, pool_S=self.pool_S, pool_st=self.pool_st,
                    pool_pad=self.pool_pad, PH=self.PH, PW=self.PW, add_en=int(self.add_src >= 0),
                    add_relu=self.add_relu, endpool=self.endpool, endpool_mult=self.endpool_mult)


def build_plan(t: NetTables) -> List[LayerSpec]:
    """Derive the explicit per-layer plan from the k* tables.

    * data input of layer l = output of conv ``kInputLayer[l]-1`` (0 = image), or concat
      tensor ``kInputLayer[l]-NUM_CONVOLUTIONS-1`` (quantization.cpp:47-49);
    * conv-stride-2 layers keep even rows/cols of the stride-1 output, i.e. are ordinary
      strided convolutions (sequencer.cl:116, pool_tail.cl:91-160);
    * residual source = the tensor most recently written to the DDR page the layer reads
      (kDDRReadBase / kDDRWriteBase / kDDRWriteEnable, feature_writer.cl:88-137).
    """
    t.validate()
    n = t.num_layer
    nconv = int(t["NUM_CONVOLUTIONS"])
    dil = t.get("xDilation", [1] * n)
    epm = t.get("xEndPoolMult", [669] * n)
    xres = t.get("xResidualSrc", None)
    page_owner: Dict[int, int] = {}
    plan: List[LayerSpec] = []
    for l in range(n):
        il = t["kInputLayer"][l]
        if il == 0:
            src = -1
        elif il <= nconv:
            src = il - 1
        else:
            src = -(il - nconv - 1 + 2)
        k, s = t["kFilterSize"][l], t["kConvStride"][l]
        H, W = t["kInputHeight"][l], t["kInputWidth"][l]
        ph, pw = t["kPadHeight"][l], t["kPadWidth"][l]
        d = dil[l]
        OH = (H + 2 * ph - d * (k - 1) - 1) // s + 1
        OW = (W + 2 * pw - d * (k - 1) - 1) // s + 1
        ipool = int(t["kIpoolEnable"][l])
        if ipool:
            OH, OW = H, W
        pool_en = int(t["kPoolEnable"][l])
        PH, PW = (t["kPoolOutputHeight"][l], t["kPoolOutputWidth"][l]) if pool_en else (OH, OW)
        if not pool_en and (t["kPoolOutputHeight"][l], t["kPoolOutputWidth"][l]) != (OH, OW) and not ipool:
            raise ConfigError(f"layer {l}: table output {t['kPoolOutputHeight'][l]}x{t['kPoolOutputWidth'][l]} "
                              f"!= computed {OH}x{OW}")
        add_src = -1
        if t["kAdditionEnable"][l]:
            if xres is not None and xres[l] >= 0:
                add_src = xres[l]
            else:
                add_src = page_owner.get(t["kDDRReadBase"][l], -1)
                if add_src < 0:
                    raise ConfigError(f"layer {l}: residual page {t['kDDRReadBase'][l]} has no writer")
        if t["kDDRWriteEnable"][l]:
            page_owner[t["kDDRWriteBase"][l]] = l
        C = t["kInputChannels"][l]
        first_rewrite = (l == 0 and t.get("xConv1Rewrite", 0))
        plan.append(LayerSpec(
            index=l, src=src, q_in_row=il, C=C, H=H, W=W, N=t["kOutputChannels"][l], k=k, stride=s,
            pad_h=ph, pad_w=pw, dil=d, OH=OH, OW=OW, bias_en=int(t["kBiasEnable"][l]),
            bn_en=int(t["kBnEnable"][l]), relu=int(t["kReluEnable"][l]), ipool=ipool,
            pool_en=pool_en, pool_S=t["kPoolWindow"][l] if pool_en else 3,
            pool_st=2 if t["kPoolStride2"][l] else 1, pool_pad=t["kPoolPad"][l], PH=PH, PW=PW,
            add_src=add_src, add_relu=int(t["kAdditionReluEnable"][l]), endpool=int(t["kEndPoolEnable"][l]),
            endpool_mult=epm[l], concat=t["kConcatLayer"][l] if t["kBranchTail"][l] else -1,
            n_start=t["kNStart"][l],
            model_C=int(t["INPUT_IMAGE_C"]) if l == 0 else C,
            model_k=int(t["FIRST_FILTER_SIZE"]) if l == 0 else k))
    return plan


def model_float_count(t: NetTables) -> int:
    """Number of float32 values LoadModel reads (model_loader.cpp:139-213)."""
    total = 0
    for L in build_plan(t):
        if not L.ipool:
            total += L.N * L.model_C * L.model_k * L.model_k
        elif L.ipool == 2:
            total += L.N                       # L2Norm row: one float scale per channel
        if L.bias_en:
            total += L.N
        if L.bn_en:
            total += 4 * L.N + 1
    return total


def q_value_count(t: NetTables) -> int:
    """Number of ints Quantization() reads from the Q file (quantization.cpp:36-53)."""
    return 3 + sum(L.N for L in build_plan(t) if L.ipool != 1)


# ----------------------------------------------------------------------------------
# Builders
# ----------------------------------------------------------------------------------
class _B:
    def __init__(self, name: str, image=(3, 224, 224), first_filter=3, rewrite=0):
        self.name = name
        self.rows: List[dict] = []
        self.image = image
        self.first_filter = first_filter
        self.rewrite = rewrite

    def conv(self, src, C, H, W, N, k, stride=1, pad=0, relu=1, bn=1, bias=0, pool=None, add=-1,
             add_relu=0, endpool=0, dil=1, endpool_hw=49, cat=None):
        """src: a row index, -1 for the image, or ("C", k) for concat tensor k.  cat = (k, n0, n1): this row is a branch tail that writes
        channels n0 .. n1 - 1 of concat tensor k (kBranchTail / kConcatLayer / kNStart / kNEnd)."""
        self.rows.append(dict(src=src, C=C, H=H, W=W, N=N, k=k, stride=stride, pad=pad, relu=relu, bn=bn,
                              bias=bias, pool=pool, add=add, add_relu=add_relu, endpool=endpool, dil=dil,
                              endpool_hw=endpool_hw, ipool=0, cat=cat))
        return len(self.rows) - 1

    def pool_only(self, src, C, H, W, pool):
        """An independent pooling row (kIpoolEnable, as GoogLeNet's inception pools): no filter, no Q row of its own."""
        self.rows.append(dict(src=src, C=C, H=H, W=W, N=C, k=1, stride=1, pad=0, relu=0, bn=0, bias=0, pool=pool,
                              add=-1, add_relu=0, endpool=0, dil=1, endpool_hw=49, ipool=1, cat=None))
        return len(self.rows) - 1

    def l2norm(self, src, C, H, W):
        """An independent L2Norm row (kIpoolEnable = 2, this build's extension for SSD's conv4_3 branch, SSD.py:46-47 /
        l2norm.py:19-24): per-pixel x / ||x||_2 times a per-channel float weight, requantised with its own Q row.  No
        filter; the model stream carries its C float weights."""
        self.rows.append(dict(src=src, C=C, H=H, W=W, N=C, k=1, stride=1, pad=0, relu=0, bn=0, bias=0, pool=None,
                              add=-1, add_relu=0, endpool=0, dil=1, endpool_hw=49, ipool=2, cat=None))
        return len(self.rows) - 1

    def tables(self) -> NetTables:
        n = len(self.rows)
        t = _blank_tables(n)
        n_concat = 1 + max([r["cat"][0] for r in self.rows if r.get("cat")], default=-1)
        t.update(INPUT_IMAGE_C=self.image[0], INPUT_IMAGE_H=self.image[1], INPUT_IMAGE_W=self.image[2],
                 FIRST_FILTER_SIZE=self.first_filter, NUM_Q_LAYERS=n + 1 + n_concat, xConv1Rewrite=self.rewrite,
                 MAX_OUT_CHANNEL=max([r["N"] for r in self.rows] + [r["cat"][2] for r in self.rows if r.get("cat")]), xName=self.name)
        if n_concat:
            t.update(xNumConcat=n_concat)
        for i, r in enumerate(self.rows):
            k, s, pad, d = r["k"], r["stride"], r["pad"], r["dil"]
            oh1 = r["H"] + 2 * pad - d * (k - 1); ow1 = r["W"] + 2 * pad - d * (k - 1)
            t["kFilterSize"][i] = k; t["kPadWidth"][i] = t["kPadHeight"][i] = pad
            t["kInputWidth"][i] = r["W"]; t["kInputHeight"][i] = r["H"]
            t["kOutputWidth"][i] = ow1; t["kOutputHeight"][i] = oh1
            t["kInputChannels"][i] = r["C"]; t["kOutputChannels"][i] = r["N"]; t["kConvStride"][i] = s
            t["kBiasEnable"][i] = r["bias"]; t["kBnEnable"][i] = r["bn"]; t["kReluEnable"][i] = r["relu"]
            t["xDilation"][i] = d; t["kNEnd"][i] = r["N"]
            t["kInputLayer"][i] = (n + 1 + r["src"][1]) if isinstance(r["src"], tuple) else r["src"] + 1
            if r.get("cat"):
                cid, n0, n1 = r["cat"]
                t["kBranchTail"][i] = 1; t["kConcatLayer"][i] = cid; t["kNStart"][i] = n0; t["kNEnd"][i] = n1
            t["kIpoolEnable"][i] = r.get("ipool", 0)
            oh, ow = (oh1 - 1) // s + 1, (ow1 - 1) // s + 1
            t["kPoolWindow"][i] = 3
            if r["pool"]:
                S, pst, ppad, PH, PW = r["pool"]
                t["kPoolEnable"][i] = 1; t["kPoolWindow"][i] = S; t["kPoolStride2"][i] = int(pst == 2)
                t["kPoolPad"][i] = ppad; t["kPoolOutputHeight"][i] = PH; t["kPoolOutputWidth"][i] = PW
            else:
                t["kPoolOutputHeight"][i] = oh; t["kPoolOutputWidth"][i] = ow
            if r["add"] >= 0:
                t["kAdditionEnable"][i] = 1; t["kAdditionReluEnable"][i] = r["add_relu"]
                t["xResidualSrc"][i] = r["add"]
            if r["endpool"]:
                t["kEndPoolEnable"][i] = 1
                t["xEndPoolMult"][i] = 669 if r["endpool_hw"] == 49 else int(round(32768.0 / r["endpool_hw"]))
        _assign_ddr_pages(t)
        return t.validate()


def resnet50_tables() -> NetTables:
    """The 54-row ResNet50 program of resnet50.h (conv1 in its 27-channel 3x3 form,
    then per block branch1 (if any), branch2a/b/c(+add+relu), end pool, fc1000)."""
    b = _B("resnet50", image=(3, 224, 224), first_filter=7, rewrite=1)
    cur = b.conv(-1, 27, 114, 114, 64, 3, 1, 0, relu=1, pool=(3, 2, 1, 56, 56))
    H = 56
    cin = 64
    for stage, (mid, nblk) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
        out = mid * 4
        for blk in range(nblk):
            s = 2 if (blk == 0 and stage > 0) else 1
            Ho = H // s
            last = (stage == 3 and blk == nblk - 1)
            if blk == 0:
                sc = b.conv(cur, cin, H, H, out, 1, s, 0, relu=0)
            else:
                sc = cur
            a = b.conv(cur, cin, H, H, mid, 1, 1, 0, relu=1)
            bb = b.conv(a, mid, H, H, mid, 3, s, 1, relu=1)
            cur = b.conv(bb, mid, Ho, Ho, out, 1, 1, 0, relu=0, add=sc, add_relu=1, endpool=int(last))
            cin, H = out, Ho
    b.conv(cur, 2048, 1, 1, 1000, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


def vgg16_tables(image_hw: int = 224, num_classes: int = 1000, with_fc: bool = True) -> NetTables:
    """VGG16 (TransForm_Kit/Quantization/models/SSD/SSD.py:89-113 base; bias convs, no BN),
    2x2/s2 max pools fused into the preceding conv; the three FC layers as 1x1 convs on the
    flattened 512x7x7 map (FC0 as a 7x7 valid conv)."""
    b = _B("vgg16", image=(3, image_hw, image_hw), first_filter=3)
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    cur, C, H = -1, 3, image_hw
    i = 0
    while i < len(cfg):
        N = cfg[i]
        pool = None
        Ho = H
        if i + 1 < len(cfg) and cfg[i + 1] == "M":
            Ho = H // 2
            pool = (2, 2, 0, Ho, Ho)
            i += 1
        cur = b.conv(cur, C, H, H, N, 3, 1, 1, relu=1, bn=0, bias=1, pool=pool)
        C, H = N, Ho
        i += 1
    if with_fc:
        cur = b.conv(cur, 512, H, H, 4096, H, 1, 0, relu=1, bn=0, bias=1)
        cur = b.conv(cur, 4096, 1, 1, 4096, 1, 1, 0, relu=1, bn=0, bias=1)
        b.conv(cur, 4096, 1, 1, num_classes, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()


def ssd300_tables(image_hw: int = 300, num_classes: int = 21, width_div: int = 1, l2norm: bool = True) -> NetTables:
    """The integer part of SSD300-VGG (TransForm_Kit/Quantization/models/SSD/SSD.py:34-77, 89-151; SURVEY.md section 8f
    rank 4) as one table program: VGG16 base with the ceil-mode third pool ('C': 75 -> 38), pool5 = 3x3 / stride 1 /
    pad 1 fused into conv5_3, conv6 = 3x3 pad 6 dilation 6, conv7 = 1x1, the four extra blocks (1x1 then 3x3 with
    stride 2 / pad 1 twice, then unpadded 3x3 twice), and the twelve multibox head convolutions (3x3 pad 1, no
    ReLU) reading their six source maps: conv4_3 through its L2Norm row (SSD.py:46-47: per-pixel x / ||x|| times a
    per-channel weight, requantised with its own Q row -- an "independent" row like the pooling rows, kIpoolEnable = 2),
    conv7, conv8_2, conv9_2, conv10_2, conv11_2.  Rows are ordered trunk
    first, heads last (loc then conf per source); every head row's output is a network output
    (read with tf2_net_read_layer).  `width_div` divides all channel counts (small test nets)."""
    d = max(1, width_div)
    b = _B("ssd300", image=(3, image_hw, image_hw), first_filter=3)
    cfgv = [64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512]
    cur, C, H = -1, 3, image_hw
    sources = []
    i = 0
    conv_idx = 0
    while i < len(cfgv):
        N = cfgv[i] // d
        pool, Ho = None, H
        is_conv4_3 = conv_idx == 9
        if i + 1 < len(cfgv) and cfgv[i + 1] in ("M", "C") and not is_conv4_3:
            Ho = H // 2 if cfgv[i + 1] == "M" else -(-H // 2)          # ceil mode: zero-extended window at the edge
            pool = (2, 2, 0, Ho, Ho)
            i += 1
        elif i == len(cfgv) - 1:
            pool = (3, 1, 1, H, H)                                       # pool5: 3x3, stride 1, pad 1
        cur = b.conv(cur, C, H, H, N, 3, 1, 1, relu=1, bn=0, bias=1, pool=pool)
        conv_idx += 1
        if is_conv4_3:
            # conv4_3 feeds its multibox heads BEFORE pool4 and THROUGH L2Norm (SSD.py:46-47): an L2Norm row of its own,
            # and the pool as an independent pooling row
            sources.append((b.l2norm(cur, N, H, H) if l2norm else cur, N, H))
            Ho = H // 2
            cur = b.pool_only(cur, N, H, H, (2, 2, 0, Ho, Ho))
            i += 1
        C, H = N, Ho
        i += 1
    cur = b.conv(cur, C, H, H, 1024 // d, 3, 1, 6, relu=1, bn=0, bias=1, dil=6)      # conv6
    cur = b.conv(cur, 1024 // d, H, H, 1024 // d, 1, 1, 0, relu=1, bn=0, bias=1)      # conv7
    C = 1024 // d
    sources.append((cur, C, H))
    for mid, out, k2, s2, p2 in ((256, 512, 3, 2, 1), (128, 256, 3, 2, 1), (128, 256, 3, 1, 0), (128, 256, 3, 1, 0)):
        cur = b.conv(cur, C, H, H, mid // d, 1, 1, 0, relu=1, bn=0, bias=1)
        Ho = (H + 2 * p2 - 3) // s2 + 1
        cur = b.conv(cur, mid // d, H, H, out // d, k2, s2, p2, relu=1, bn=0, bias=1)
        C, H = out // d, Ho
        sources.append((cur, C, H))
    mbox = [4, 6, 6, 6, 4, 4]
    for (src, Cs, Hs), nb in zip(sources, mbox):
        b.conv(src, Cs, Hs, Hs, nb * 4, 3, 1, 1, relu=0, bn=0, bias=1)                  # loc head
        b.conv(src, Cs, Hs, Hs, nb * num_classes, 3, 1, 1, relu=0, bn=0, bias=1)        # conf head
    return b.tables()


SQUEEZENET11_FIRES = ((16, 64, False), (16, 64, True), (32, 128, False), (32, 128, True),
                      (48, 192, False), (48, 192, False), (64, 256, False), (64, 256, False))


def squeezenet11_tables(image_hw: int = 227) -> NetTables:
    """SqueezeNet 1.1 as in TransForm_Kit/Quantization/models/SqueezeNet/SqueezeNet.py:17-113
    (conv+BN everywhere except final_conv; fire = squeeze1x1 -> expand1x1 || expand3x3,
    ceil-mode 3x3/s2 pools; final_conv + global average without ReLU; fc 1000 -> 128 + BN as a 1x1 row on a
    signed input, which the engine runs on the shift-accumulate kernel).  Concats use the reference's branch-tail encoding
    (kBranchTail/kConcatLayer/kNStart/kNEnd); a pool after a concat is
    distributed onto the branch tails as GoogLeNet's header does."""
    return fire_net_tables(image_hw, (64, 3, 2, 0, True), SQUEEZENET11_FIRES, 1000, 128, "squeezenet1_1")


def fire_net_tables(image_hw: int, conv1, fires, classes: int, fc_out: int, name: str) -> NetTables:
    """A SqueezeNet-shaped program: conv1 = (N, k, stride, pad, pooled) on the 3-channel image, fire modules (squeeze, expand, pool after),
    final conv to `classes` + global average, fc to `fc_out` (squeezenet11_tables is the 1.1 instance; tests draw others)."""
    rows: List[dict] = []
    n_concat = 0

    def add(**kw):
        rows.append(kw); return len(rows) - 1

    def ceil_pool(h):
        return -(-(h - 3) // 2) + 1

    N1, k1, s1, p1, pooled1 = conv1
    H1 = (image_hw + 2 * p1 - k1) // s1 + 1
    P1 = ceil_pool(H1) if pooled1 else H1
    add(src=("L", -1), C=3, H=image_hw, N=N1, k=k1, stride=s1, pad=p1, pool=(3, 2, 0, P1, P1) if pooled1 else None, cat=None)
    cur, C, H = ("L", 0), N1, P1
    for sq, ex, pool_after in fires:
        s = add(src=cur, C=C, H=H, N=sq, k=1, stride=1, pad=0, pool=None, cat=None)
        Ho = ceil_pool(H) if pool_after else H
        pool = (3, 2, 0, Ho, Ho) if pool_after else None
        add(src=("L", s), C=sq, H=H, N=ex, k=1, stride=1, pad=0, pool=pool, cat=(n_concat, 0, ex))
        add(src=("L", s), C=sq, H=H, N=ex, k=3, stride=1, pad=1, pool=pool, cat=(n_concat, ex, 2 * ex))
        cur, C, H = ("C", n_concat), 2 * ex, Ho
        n_concat += 1
    # final_conv (bias, no BN, NO ReLU: SqueezeNet.py:101-102) + global average, then fc 1000 -> 128 (no bias) + BN (:103-108)
    fin = add(src=cur, C=C, H=H, N=classes, k=1, stride=1, pad=0, pool=None, cat=None, bias=1, bn=0, relu=0, endpool=H * H)
    add(src=("L", fin), C=classes, H=1, N=fc_out, k=1, stride=1, pad=0, pool=None, cat=None, bias=0, bn=1, relu=0)
    n = len(rows)
    t = _blank_tables(n)
    t.update(INPUT_IMAGE_C=3, INPUT_IMAGE_H=image_hw, INPUT_IMAGE_W=image_hw, FIRST_FILTER_SIZE=3,
             NUM_Q_LAYERS=n + 1 + n_concat, xConv1Rewrite=0, MAX_OUT_CHANNEL=max([r["N"] for r in rows] + [r["cat"][2] for r in rows if r["cat"]]), xName=name,
             xNumConcat=n_concat)
    for i, r in enumerate(rows):
        k, s, pad = r["k"], r["stride"], r["pad"]
        oh1 = r["H"] + 2 * pad - (k - 1)
        t["kFilterSize"][i] = k; t["kPadWidth"][i] = t["kPadHeight"][i] = pad
        t["kInputWidth"][i] = t["kInputHeight"][i] = r["H"]
        t["kOutputWidth"][i] = t["kOutputHeight"][i] = oh1
        t["kInputChannels"][i] = r["C"]; t["kOutputChannels"][i] = r["N"]; t["kConvStride"][i] = s
        t["kBiasEnable"][i] = r.get("bias", 0); t["kBnEnable"][i] = r.get("bn", 1); t["kReluEnable"][i] = r.get("relu", 1)
        kind, idx = r["src"]
        t["kInputLayer"][i] = (idx + 1) if kind == "L" else (n + 1 + idx)
        oh = (oh1 - 1) // s + 1
        t["kPoolWindow"][i] = 3
        if r["pool"]:
            S, pst, ppad, PH, PW = r["pool"]
            t["kPoolEnable"][i] = 1; t["kPoolStride2"][i] = int(pst == 2); t["kPoolPad"][i] = ppad
            t["kPoolOutputHeight"][i] = PH; t["kPoolOutputWidth"][i] = PW
        else:
            t["kPoolOutputHeight"][i] = t["kPoolOutputWidth"][i] = oh
        if r["cat"]:
            cid, n0, n1 = r["cat"]
            t["kBranchTail"][i] = 1; t["kConcatLayer"][i] = cid; t["kNStart"][i] = n0; t["kNEnd"][i] = n1
        else:
            t["kNEnd"][i] = r["N"]
        if r.get("endpool"):
            t["kEndPoolEnable"][i] = 1
            t["xEndPoolMult"][i] = 669 if r["endpool"] == 49 else int(round(32768.0 / r["endpool"]))
    return t.validate()


def tiny_tables(hw: int = 12, c0: int = 3, widths=(16, 32), classes: int = 10) -> NetTables:
    """A few-layer residual network exercising every post-op (pool, stride-2 conv,
    residual add, global average, fc) at sizes the CPU oracle finishes instantly."""
    b = _B("tiny", image=(c0, hw, hw), first_filter=3)
    w0, w1 = widths
    a = b.conv(-1, c0, hw, hw, w0, 3, 1, 1, relu=1, pool=(3, 2, 1, hw // 2, hw // 2))
    h = hw // 2
    sc = b.conv(a, w0, h, h, w1, 1, 2, 0, relu=0)
    x = b.conv(a, w0, h, h, w0, 1, 1, 0, relu=1)
    x = b.conv(x, w0, h, h, w0, 3, 2, 1, relu=1)
    h2 = (h - 1) // 2 + 1
    x = b.conv(x, w0, h2, h2, w1, 1, 1, 0, relu=0, add=sc, add_relu=1)
    y = b.conv(x, w1, h2, h2, w0, 1, 1, 0, relu=1)
    y = b.conv(y, w0, h2, h2, w0, 3, 1, 1, relu=1)
    y = b.conv(y, w0, h2, h2, w1, 1, 1, 0, relu=0, add=x, add_relu=1, endpool=1, endpool_hw=h2 * h2)
    b.conv(y, w1, 1, 1, classes, 1, 1, 0, relu=0, bn=0, bias=1)
    return b.tables()
