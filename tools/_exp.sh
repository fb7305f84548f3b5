#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "group_launches" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_configs.py -q -x -k "graph_replay" 2>&1 | tail -2
timeout 300 python tools/bgroup_stress.py 150 32,48,5,64,32 2>&1 | tail -5
for c in 1 5 1 5; do
  echo -n "chain=$c: "; TF2_AMD_BGROUP_CHAIN=$c timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'])"
done
