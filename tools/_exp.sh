#!/bin/bash
mkdir -p gpurun_out
run() { echo -n "$1: "; timeout 300 python bench.py --no-cpu --steps 40 --warmup 5 --extra-batches "" $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['xcd_partitions'])"; }
for r in 1 2; do
run default ""
run five_parts "--inflight 5 --partition 2"
run five_plain "--inflight 5 --partition 0"
run six_parts "--inflight 6 --partition 2"
run three_parts "--inflight 3 --partition 2"
done
