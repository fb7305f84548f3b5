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
    # TOP LEVEL = the plan the timed region runs (batches in flight): achieved = algorithmic bytes per step / ms_per_step, its own launch
    # count, kernel time and classes, named by the library; at least eight distinct input buffers rotate through the timed region
    assert d["config"]["input_buffers_rotated"] >= 8 and d["config"]["batches_in_flight"] == 4
    assert r["kernel_us_per_step"] > 0 and r["kernel"]["kernel"].startswith("conv_") and "conv_bband_kernel" in r["kernels"]
    assert abs(r["achieved"] - d["hbm"]["algorithmic_gbps"]) / r["achieved"] < 0.05 and r["overlap_factor"] > 1.0
    assert r["kernel"]["frac_hbm_peak"] > 0 and r["kernel"]["frac_int8_peak"] > 0 and r["kernel"]["rows"][0] == r["kernel"]["first_row"]
    assert abs(sum(v["share_of_kernel_time"] for v in d["per_layer_class"].values()) - 1.0) < 0.01
    # ... the one-batch-at-a-time plan (group launches) nested under it
    o = r["one_batch"]
    assert o["launches_per_step"] < r["launches_per_step"] and o["kernel_us_per_step"] > 0 and "conv_bgroup_kernel" in o["kernels"]
    assert d["per_layer_class_one_batch"]
    # the committed rocprofv3 summaries of the same workload (profiles/r06_rocprof_b32[_conc1]_summary.json, tools/round_evidence.sh)
    # agree with the live HIP-event figures
    for blk in (r, o):
        if blk.get("kernel_us_per_step_rocprof"):
            assert abs(blk["kernel_us_per_step_rocprof"] - blk["kernel_us_per_step"]) / blk["kernel_us_per_step"] < 0.10, (blk["kernel_us_per_step_rocprof"], blk["kernel_us_per_step"])
    # the K-step region repeated (round 6): `value` is the first repetition, min / median / max beside it
    rp = d["repeats"]
    assert len(rp["values"]) >= 5 and rp["values"][0] == d["value"] and rp["min"] <= rp["median"] <= rp["max"]
    assert d["cold_start"]["value"] > 0 and d["weight_broadcast"]["ms"] is None and d["per_rank"] is None        # one GPU: no collective ran


@pytest.mark.gpu
def test_bench_other_network_same_line():
    """bench.py --net: the other BASELINE.json configurations through the same measurement (roofline + cpu_baseline in the line)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--net", "squeezenet", "--steps", "6", "--warmup", "2", "--cpu-seconds", "2",
                          "--extra-batches", ""], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert "SqueezeNet" in d["metric"] and "227x227" in d["config"]["workload"] and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["roofline"]["one_batch"]["one_at_a_time"]["frac"] > 0 and d["cpu_baseline"]["parity_with_gpu_logits"] is True
