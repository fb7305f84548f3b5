#!/bin/bash
# bench.py over batch sizes x stream counts (run on the GPU box)
for b in ${BATCHES:-32 64}; do for st in ${STREAMS:-1 2 4}; do
  python bench.py --no-cpu --steps 30 --warmup 5 --batch $b --streams $st --extra-batches "" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b streams $st', d['value'], d['ms_per_step'])"
done; done
