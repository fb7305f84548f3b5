#!/bin/bash
# Round 2, first GPU session: parity tests, smoke, bench lines (default / serial variants / mode 1), rocprofv3 kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; tail -c 600 $O/bench_default.log
for v in "--inflight 1 --split 2" "--inflight 1 --split 4" "--inflight 1 --graph 1" "--inflight 1 --graph 1 --split 2" "--inflight 3 --split 2" "--inflight 2" "--inflight 4"; do
  n=$(echo $v | tr -d ' -'); timeout 300 python bench.py --no-cpu --steps 60 --extra-batches "" $v > $O/bench_$n.log 2>&1
  echo "$v: $(tail -1 $O/bench_$n.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["images_per_s_one_batch_at_a_time"], d["latency_batch1"])' 2>&1 | tail -1)"
done
timeout 300 python bench.py --no-cpu --mode 1 --extra-batches "" --steps 30 > $O/bench_mode1.log 2>&1; tail -1 $O/bench_mode1.log > $O/bench_mode1.json; tail -c 300 $O/bench_mode1.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o ks -- python $R/bench.py --no-cpu --inflight 1 --steps 20 --warmup 5 --extra-batches "" > $O/rocprof_serial.log 2>&1
find /tmp/prof_serial -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b32_serial.csv \;
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fl3 -o kt -- python $R/bench.py --no-cpu --inflight 3 --steps 20 --warmup 5 --extra-batches "" > $O/rocprof_fl3.log 2>&1
find /tmp/prof_fl3 -name "*kernel_trace.csv" -exec cp {} /tmp/kt_fl3.csv \;
python $R/tools/trace_overlap.py /tmp/kt_fl3.csv 0.6 > $O/overlap_inflight3.json 2>&1; cat $O/overlap_inflight3.json
head -c 400000 /tmp/kt_fl3.csv | tail -c 150000 > $O/kernel_trace_inflight3_excerpt.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m1 -o ks -- python $R/bench.py --no-cpu --mode 1 --inflight 1 --steps 10 --warmup 3 --extra-batches "" > $O/rocprof_mode1.log 2>&1
find /tmp/prof_m1 -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b32_mode1.csv \;
head -8 $O/rocprof_kernel_stats_b32_serial.csv; head -5 $O/rocprof_kernel_stats_b32_mode1.csv
