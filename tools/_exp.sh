#!/bin/bash
mkdir -p gpurun_out/evidence
timeout 300 python -m pytest tests/test_bench_contract.py -q -x 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/evidence/bench_default.log 2>&1; tail -1 gpurun_out/evidence/bench_default.log > gpurun_out/evidence/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/evidence/bench_default.json').read())
print(d['value'], d['cold_start']['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'], list(d['per_layer_class'].keys()))
PY
timeout 300 python tools/bgroup_stress.py 400 32,48,5,64,32 2>&1 | tail -5
