#!/usr/bin/env python3
"""Static check of the hand-written `s_waitcnt vmcnt(N)` waits of the LDS-DMA kernels, on the COMPILED code (csrc/vm_track.h).

conv_bband.hip / conv_c3.hip issue their LDS-DMAs (`global_load_lds_dwordx4`) and the waits for them as inline assembly.  The VM
counter retires in order, so a wait `vmcnt(N)` in front of a barrier covers the wave's last group of DMAs exactly when, on EVERY
path of the control-flow graph that reaches the wait, at least N vector-memory LOADS were issued behind the last DMA instruction.
This tool takes the device assembly hipcc writes for the shipped sources with the shipped flags (`make -C tf2_amd/csrc isa` ->
csrc/build/isa/*.s; inline assembly is bracketed by ;;#ASMSTART / ;;#ASMEND there), builds the CFG of every kernel and runs a
forward dataflow with min-merge:

    state  = VM loads issued since the last DMA instruction (capped at 64; 64 = no DMA on this path yet)
    DMA    -> 0        VM load -> state + 1        hand-written vmcnt(N), N > 0:  require state >= N

Only LOADS are counted (they execute under the wave's full exec mask in these kernels; a store under an empty exec mask may
not enter the counter, and no wait relies on one).  Kernels that keep DMAs of several groups in flight (`conv_c3_w9_kernel`: three
tiles ahead) choose N at run time from what was really issued (vm_wait_groups): there the tool checks the structure instead --
every hand-written N is a multiple of the DMAs one produce() issues per wave, and the ladder ends in vmcnt(0).

  python tools/vmcnt_check.py [file.s ...]      exit status 1 on any violation"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tf2_amd", "csrc")
ISA_DIR = os.path.join(CSRC, "build", "isa")
SOURCES = ("conv_bband", "conv_c3")
INF = 64

VM_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load|global_atomic\w*_rtn|buffer_atomic\w*_rtn)")
BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\.L\w+)")
LABEL = re.compile(r"^(\.L\w+):")
KERNEL = re.compile(r"^(_Z\w+):")
VMCNT = re.compile(r"^s_waitcnt\s+.*vmcnt\((\d+)\)")


def build_isa(force=False):
    """device assembly of the DMA kernels with the library's own flags (Makefile target `isa`)"""
    subprocess.check_call(["make", "-s", "-C", CSRC, "isa"] + (["-B"] if force else []), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(ISA_DIR, s + ".s") for s in SOURCES]


def parse_kernels(path):
    """{kernel symbol: [(text, in_asm)]} -- instructions and labels of every kernel function, in layout order"""
    kernels, cur, in_asm = {}, None, False
    for raw in open(path):
        s = raw.strip()
        if ";;#ASMSTART" in s:
            in_asm = True
            continue
        if ";;#ASMEND" in s:
            in_asm = False
            continue
        m = KERNEL.match(s)
        if m and not s.startswith(".L"):
            cur = kernels.setdefault(m.group(1), [])
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None or not s:
            continue
        s = s.split(";")[0].strip()
        if not s or (s.startswith(".") and not LABEL.match(s)):
            continue
        cur.append((s, in_asm))
    return kernels


def analyse(items):
    """dataflow over one kernel; returns (waits, n_dma) with waits = [(N, min state at the wait, index)]"""
    # basic blocks: leaders = labels and instructions behind a branch
    label_at = {}
    for i, (s, _) in enumerate(items):
        m = LABEL.match(s)
        if m:
            label_at[m.group(1)] = i
    n = len(items)
    succ = [[] for _ in range(n)]
    for i, (s, _) in enumerate(items):
        m = BRANCH.match(s)
        if m:
            tgt = label_at.get(m.group(2))
            if tgt is not None:
                succ[i].append(tgt)
            if m.group(1) != "s_branch" and i + 1 < n:
                succ[i].append(i + 1)
        elif s.startswith("s_endpgm") or s.startswith("s_setpc") or s.startswith("s_trap"):
            pass
        elif i + 1 < n:
            succ[i].append(i + 1)
    state_in = [None] * n          # min over paths of the state BEFORE instruction i (None: unreached)
    state_in[0] = INF
    work = [0]
    n_dma = 0
    is_dma = [a and s.startswith("global_load_lds") for s, a in items]
    n_dma = sum(is_dma)
    while work:
        i = work.pop()
        st = state_in[i]
        s, a = items[i]
        if is_dma[i]:
            out = 0
        elif VM_LOAD.match(s) and not s.startswith("global_load_lds"):
            out = min(st + 1, INF)
        else:
            out = st
        for j in succ[i]:
            if state_in[j] is None or out < state_in[j]:
                state_in[j] = out
                work.append(j)
    waits = []
    for i, (s, a) in enumerate(items):
        m = VMCNT.match(s)
        if a and m and state_in[i] is not None:
            waits.append((int(m.group(1)), state_in[i], i))
    return waits, n_dma


def check_file(path, verbose=True):
    bad = 0
    report = []
    for name, items in parse_kernels(path).items():
        waits, n_dma = analyse(items)
        if not n_dma:
            continue
        short = name[:100]
        if "conv_c3_w9" in name:
            # run-time ladder (vm_wait_groups<NG, 2>): N in {2 NG, NG, 0} with NG = the DMAs one produce() issues per wave (one per
            # 64-pixel group of the halo tile: kC3HaloPx / 64 = 6; wave-uniform branches around the address arithmetic separate the
            # DMA instructions of a call site, none skips a DMA: their total is a multiple of NG)
            ns = sorted({w[0] for w in waits if w[0] > 0})
            ng = ns[0] if ns else 0
            ok = ng == 6 and ns == [ng, 2 * ng] and n_dma % ng == 0 and any(w[0] == 0 for w in waits)
            report.append(f"{'ok ' if ok else 'BAD'} {short}: run-time ladder vmcnt{[0] + ns}, {n_dma} DMA instructions = {n_dma // max(ng, 1)} produce() sites x {ng}")
            bad += 0 if ok else 1
            continue
        counted = [w for w in waits if w[0] > 0]
        for N, st, i in counted:
            ok = st >= N
            bad += 0 if ok else 1
            report.append(f"{'ok ' if ok else 'BAD'} {short}: vmcnt({N}) at instruction {i}: >= {st if st < INF else 'no DMA before it: any'} VM loads behind the last DMA on every path")
        if not counted:
            report.append(f"ok  {short}: {n_dma} DMA instructions, every hand-written wait is vmcnt(0)")
    if verbose:
        print("\n".join(report))
    return bad, report


def main(argv):
    files = argv or build_isa()
    bad = 0
    for f in files:
        print(f"== {os.path.relpath(f, ROOT)}")
        b, _ = check_file(f)
        bad += b
    print(f"{bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
