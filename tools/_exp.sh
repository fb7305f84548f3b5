#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/gputests.log 2>&1; tail -2 $R/gpurun_out/gputests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats_32
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_32 -o ks -- python $R/tools/steps_only.py --batch 32 --steps 40 --meta $O/steps_b32.json > $O/rocprof_stats_b32.log 2>&1
find /tmp/prof_stats_32 -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b32.csv \;
python $R/tools/rocprof_summary.py $O/rocprof_kernel_stats_b32.csv $O/steps_b32.json $O/rocprof_b32_summary.json
cd $R
cp $O/rocprof_b32_summary.json profiles/r03_rocprof_b32_summary.json
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 200 python tools/layer_times.py --batch 32 --stamps > $O/layer_times_b32.txt 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/evidence/bench_default.json').read())
print(d['value'], d['cold_start']['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'], d['roofline']['kernel_us_per_step'], d['roofline']['kernel_us_per_step_rocprof'], d['roofline']['launches_per_step'], d['latency_batch1']['us_per_image'], d['images_per_s_by_batch'], d['cpu_baseline']['value'])
PY
