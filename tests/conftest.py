import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# one hardware queue per in-flight stream (bench.py): read by the HIP runtime when it initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side prerequisites: the oracle (plain C) and the C-ABI library (host entry points work
    # without a GPU; hipcc cross-compiles gfx950).  Built once per session if stale/missing.
    from oracle import oracle as O
    O.build()
    from tf2_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def set_opts(monkeypatch, **kw):
    """Set / change / remove (value None) options of the library's ONE option string TF2_AMD_OPTS (csrc/opts.h), cumulatively within a
    test; TF2_AMD_TEST=1 admits the test-only ones (forced kernels, disabled proofs, thresholds).  Takes effect at the next
    tf2_net_create / tf2_net_reload_options."""
    from tf2_amd._lib import parse_opts
    cur = parse_opts(os.environ.get("TF2_AMD_OPTS", ""))
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    monkeypatch.setenv("TF2_AMD_TEST", "1")
    monkeypatch.setenv("TF2_AMD_OPTS", ",".join(f"{k}={v}" for k, v in cur.items()))
