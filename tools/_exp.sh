#!/bin/bash
mkdir -p gpurun_out
for r in 1 2 3 4; do
for f in 0 1; do
  echo -n "feeder=$f: "; timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --extra-batches "" --feeder $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('images_per_s_one_batch_at_a_time'), d['roofline']['frac'])"
done
done
