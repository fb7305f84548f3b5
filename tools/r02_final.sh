cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'], d['cpu_baseline'])"
