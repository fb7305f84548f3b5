#!/bin/bash
# round 5, GPU call 2: parity of fc4 / RNN / im2col rows, then bench lines (no CPU leg)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call2; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for NET in resnet50 vgg16 ssd300 squeezenet; do
  timeout 300 python bench.py --net $NET --no-cpu --steps 40 --warmup 5 --extra-batches "" > $O/bench_$NET.log 2>&1; tail -1 $O/bench_$NET.log > $O/bench_$NET.json
  python - <<EOF
import json
try:
    d=json.load(open("$O/bench_$NET.json")); r=d["roofline"]
    print("$NET", d["value"], "cold", d["cold_start"]["value"], "one-batch", d["images_per_s_one_batch_at_a_time"], "kernel", r["kernel"]["kernel"][:60], r["kernel"]["us"], "kus", r["kernel_us_per_step"], r["one_batch"]["kernel_us_per_step"])
except Exception as e: print("$NET failed", e)
EOF
done
timeout 120 python tools/layer_times.py --batch 32 > $O/layer_times_b32.txt 2>&1; tail -5 $O/layer_times_b32.txt
