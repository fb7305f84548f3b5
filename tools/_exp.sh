#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/gputests.log 2>&1; tail -3 $R/gpurun_out/gputests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" > $O/bench_s100.log 2>&1; tail -1 $O/bench_s100.log > $O/bench_s100.json
timeout 300 python bench.py --no-cpu --batch 64 --extra-batches "" > $O/bench_b64.log 2>&1; tail -1 $O/bench_b64.log > $O/bench_b64.json
timeout 300 python bench.py --no-cpu --batch 1 --extra-batches "" > $O/bench_b1.log 2>&1; tail -1 $O/bench_b1.log > $O/bench_b1.json
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 --extra-batches "" --spinup-ms 0 > $O/bench_nospin.log 2>&1; tail -1 $O/bench_nospin.log > $O/bench_s20_no_spinup.json
timeout 200 python tools/short_run_timeline.py 20 4 5 0 > $O/timeline_f0.txt 2>&1
timeout 200 python tools/short_run_timeline.py 20 4 5 1 > $O/timeline_f1.txt 2>&1
timeout 100 python tools/clock_sample.py > $O/clock_sample.txt 2>&1
for f in bench_default bench_s100 bench_b64 bench_b1 bench_s20_no_spinup; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'serial', d.get('images_per_s_one_batch_at_a_time'), 'frac', d['roofline']['frac'], d['roofline'].get('kernel_us_per_step_rocprof'), d['roofline'].get('kernel_us_per_step'), 'inflight', d['roofline']['in_flight']['frac'], 'lat', (d.get('latency_batch1') or {}).get('us_per_image'), d.get('images_per_s_by_batch'), (d.get('cpu_baseline') or {}).get('value'))
PY
done
