#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/bgroup_timeline.py --layer 31 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "group_launches_of" 2>&1 | tail -2
for c in 1 2; do
  timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'])"
done
