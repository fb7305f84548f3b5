#!/bin/bash
mkdir -p gpurun_out/evidence
timeout 300 python -m pytest tests/test_bench_contract.py -q -x 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/evidence/bench_default.log 2>&1; tail -1 gpurun_out/evidence/bench_default.log > gpurun_out/evidence/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/evidence/bench_default.json').read())
print(d['value'], d['cold_start']['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'])
for k,v in d['per_layer_class'].items(): print(k, v)
PY
timeout 600 python tools/bench_configs.py > gpurun_out/evidence/other_configs.txt 2>&1; tail -4 gpurun_out/evidence/other_configs.txt
