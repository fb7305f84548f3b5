#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O; rm -f $O/exp.txt
for e in 0 2 4 0 2; do
  echo "== TF2_AMD_EXP=$e" >> $O/exp.txt
  TF2_AMD_EXP=$e timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu --extra-batches "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('in flight', d['value'], 'serial', d.get('images_per_s_one_batch_at_a_time'), 'frac', d['roofline']['frac'])" >> $O/exp.txt
done
cat $O/exp.txt
