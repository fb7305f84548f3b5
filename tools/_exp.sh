#!/bin/bash
mkdir -p gpurun_out/evidence
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/evidence/bench_default.log 2>&1; tail -1 gpurun_out/evidence/bench_default.log > gpurun_out/evidence/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/evidence/bench_default.json').read())
print(d['value'], d['ms_per_step'], d['cold_start'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'], d['config']['spinup_ms'], d['config']['hip_graph'])
PY
timeout 300 python -m pytest tests/test_bench_contract.py -q -x 2>&1 | tail -2
