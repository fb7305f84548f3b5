"""bench.py prints ONE JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-seconds", "2",
                          "--extra-batches", "1"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "int8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 32 * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 0.01
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "mfma_side" in r and d["latency_batch1"]["us_per_image"] > 0 and d["images_per_s_one_batch_at_a_time"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["parity_with_gpu_logits"] is True
    # the committed rocprofv3 summary of the same workload (profiles/rNN_rocprof_b32_summary.json, tools/round_evidence.sh) agrees
    # with the live HIP-event figure behind `roofline.frac`
    assert r["kernel_us_per_step"] > 0 and r["kernel"]["kernel"].startswith("conv_") and "conv_bgroup_kernel" in r["kernels"]
    # the plan the timed region runs (batches in flight): its own launch count, kernel time and classes, named by the library
    f = r["in_flight"]
    assert f["launches_per_step"] > r["launches_per_step"] and f["kernel_us_per_step"] > 0 and "conv_bband_kernel" in f["kernels"]
    assert abs(f["frac"] - f["achieved"] / r["peak"]) < 1e-3 and d["per_layer_class_in_flight"]
    assert abs(sum(v["share_of_kernel_time"] for v in d["per_layer_class_in_flight"].values()) - 1.0) < 0.01
    if f.get("kernel_us_per_step_rocprof"):
        assert abs(f["kernel_us_per_step_rocprof"] - f["kernel_us_per_step"]) / f["kernel_us_per_step"] < 0.10, (f["kernel_us_per_step_rocprof"], f["kernel_us_per_step"])
    if r.get("kernel_us_per_step_rocprof"):
        assert abs(r["kernel_us_per_step_rocprof"] - r["kernel_us_per_step"]) / r["kernel_us_per_step"] < 0.10, (r["kernel_us_per_step_rocprof"], r["kernel_us_per_step"])


@pytest.mark.gpu
def test_bench_other_network_same_line():
    """bench.py --net: the other BASELINE.json configurations through the same measurement (roofline + cpu_baseline in the line)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--net", "squeezenet", "--steps", "6", "--warmup", "2", "--cpu-seconds", "2",
                          "--extra-batches", ""], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert "SqueezeNet" in d["metric"] and "227x227" in d["config"]["workload"] and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["roofline"]["in_flight"]["frac"] > 0 and d["cpu_baseline"]["parity_with_gpu_logits"] is True
