#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "graph or in_flight" 2>&1 | tail -3
run() { echo -n "$1: "; timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --extra-batches "" $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['images_per_s_one_batch_at_a_time'], d['latency_batch1']['by_path'], d['config']['hip_graph'])"; }
for r in 1 2; do
run default ""
run graph0 "--graph 0"
run graph1 "--graph 1"
done
