"""The hand-written `s_waitcnt vmcnt(N)` waits of the LDS-DMA kernels, checked on the COMPILED code (csrc/vm_track.h,
tools/vmcnt_check.py): on every path of every instantiation's control-flow graph at least N vector-memory loads lie between the last
`global_load_lds_dwordx4` and the wait that covers it -- the property whose violation is bit-exact on an idle chip and wrong under load
(round 4: b53d5bf, 7173314).  Needs hipcc (device assembly of conv_bband.hip / conv_c3.hip with the library's own flags), no GPU."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import vmcnt_check  # noqa: E402


@pytest.fixture(scope="module")
def isa_files():
    return vmcnt_check.build_isa()


def test_every_counted_wait_is_covered_on_every_path(isa_files):
    n_waits = 0
    for f in isa_files:
        bad, report = vmcnt_check.check_file(f, verbose=False)
        assert bad == 0, "\n".join(r for r in report if r.startswith("BAD"))
        n_waits += len(report)
    # every instantiation the launchers can select is in the assembly: 8 band kernels, the halo-tile kernels (with / without the 2x2 pool
    # epilogue) + the one-slab kernel in both forms
    kernels = {}
    for f in isa_files:
        kernels.update(vmcnt_check.parse_kernels(f))
    assert sum("conv_bband_kernel" in k for k in kernels) == 9     # (round 6: + the 7-row band of the 28 x 28 bottleneck whose reduce AND 3x3 are two-window layers)
    # no band kernel parks a register in scratch (round 4: blocks of 1 ms once a lambda took the accumulators by reference; rounds 4-5 kept
    # the two-window / two-window bottleneck off 7-row bands because that instantiation spilled 23 registers then)
    for f in isa_files:
        txt = open(f).read()
        for m in re.finditer(r"\.name:\s+(\S*conv_bband_kernel\S*)\n\s+\.private_segment_fixed_size:\s+(\d+)", txt):
            assert int(m.group(2)) == 0, (m.group(1), m.group(2))
    assert sum("conv_c3_kernel" in k for k in kernels) >= 10 and sum("conv_c3_w9_kernel" in k for k in kernels) == 2
    assert n_waits >= 40


def test_the_checker_sees_a_wait_that_is_one_load_too_generous(isa_files):
    """the same analysis on a doctored copy: one vector-memory load removed from behind a chunk's DMAs must be reported"""
    kernels = vmcnt_check.parse_kernels(isa_files[0])
    name, items = next((k, v) for k, v in kernels.items() if "conv_bband_kernel" in k)
    waits, n_dma = vmcnt_check.analyse(items)
    tight = [w for w in waits if w[0] > 0 and w[1] == min(x[1] for x in waits if x[0] > 0)][0]
    N, st, at = tight
    # drop (st - N + 1) loads in front of that wait: the wait then allows one more than was issued
    doctored, dropped, need = list(items), 0, st - N + 1
    for i in range(at - 1, -1, -1):
        if dropped == need:
            break
        t, a = doctored[i]
        if a and t.startswith("global_load_lds"):
            break
        if vmcnt_check.VM_LOAD.match(t):
            doctored[i] = ("s_nop 0", False)
            dropped += 1
    assert dropped == need
    waits2, _ = vmcnt_check.analyse(doctored)
    assert any(w[2] == at and w[1] < w[0] for w in waits2)


def test_no_vmcnt_literal_left_in_the_dma_kernels():
    """every hand-written vmcnt wait of the DMA kernels goes through vm_track.h (vmcnt(0) drains excepted)"""
    for f in ("conv_bband.hip", "conv_c3.hip"):
        src = open(os.path.join(ROOT, "tf2_amd", "csrc", f)).read()
        lits = [m.group(1) for m in re.finditer(r"s_waitcnt[^\"]*vmcnt\((\d+)\)", src)]
        assert all(v == "0" for v in lits), (f, lits)
        assert "vm_wait<" in src


def test_the_two_block_per_cu_bottleneck_kernels_compile_without_scratch(isa_files):
    """Round 6: spilled registers are HBM traffic -- a lane's private segment is written back as dirty lines and partly re-fetched.  conv_bfirst's
    84 bytes were 17 MB written + 16 MB fetched per launch (half of the kernel's traffic in the PMC passes), conv_bneck's 52-80 bytes the 10.5-11.5 MB
    by which its WRITE_SIZE exceeded its output; without them +0.9 % and +1.0-1.6 % img/s with batches in flight (profiles/r06_experiments.txt item 18).
    The compiled listing of the shipped sources with the shipped flags: conv_bfirst (both instantiations) and the 64-channel shapes of conv_bneck
    (the ones ResNet-50 launches) stay at a private segment of 0 (one shape: three registers)."""
    isa_dir = os.path.dirname(isa_files[0])
    seg = {}
    for src in ("conv_bfirst", "conv_bneck"):
        txt = open(os.path.join(isa_dir, src + ".s")).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", txt):
            seg[m.group(1)] = int(m.group(2))
    bfirst = {k: v for k, v in seg.items() if "conv_bfirst_kernel" in k}
    assert len(bfirst) == 2 and all(v == 0 for v in bfirst.values()), bfirst
    bneck64 = {k: v for k, v in seg.items() if "conv_bneck_kernelILi2ELi4ELi2E" in k}
    assert len(bneck64) == 4 and all(v <= 12 for v in bneck64.values()) and sum(v == 0 for v in bneck64.values()) >= 2, bneck64


def test_the_short_k_pointwise_kernel_compiles_without_scratch(isa_files):
    """conv_pwk.hip: every instantiation (2 / 4 K slabs x 2 / 4 channel groups, 8 slabs x 4; one / two windows; single and pair launches) at a private segment of 0 -- its first forms
    (eight waves at 128 registers beside up to 64 resident fragment registers) parked 160-470 bytes per lane; four waves per block at <= 256 registers."""
    txt = open(os.path.join(os.path.dirname(isa_files[0]), "conv_pwk.s")).read()
    seg = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", txt)}
    mine = {k: v for k, v in seg.items() if "conv_pwk_" in k}
    assert len(mine) == 20, mine                                  # (2 / 4 slabs x 2 / 4 channel groups + 8 slabs x 4) x one / two windows, single and pair launches
    assert all(v == 0 for v in mine.values()), mine           # (the two-window K = 512 ones take 293 registers: one block per CU, launch bounds (256, 1))
