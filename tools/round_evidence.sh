#!/bin/bash
# Round evidence on the GPU box: bench line (with cpu_baseline), rocprofv3 kernel stats of the same workload one batch at a
# time (the condition under which bench.py measures the per-kernel durations behind `roofline`), PMC passes for HBM traffic,
# microbenchmarks.  Outputs under gpurun_out/evidence/ (copied to profiles/r02_* by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 300 python bench.py --no-cpu --batch 64 --extra-batches "" > $O/bench_b64.log 2>&1; tail -1 $O/bench_b64.log > $O/bench_b64.json
timeout 300 python bench.py --no-cpu --batch 1 --extra-batches "" > $O/bench_b1.log 2>&1; tail -1 $O/bench_b1.log > $O/bench_b1.json
timeout 200 python tools/layer_times.py --batch 32 --stamps > $O/layer_times_b32.txt 2>&1
timeout 200 python tools/layer_times.py --batch 1 > $O/layer_times_b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o ks -- python $R/bench.py --no-cpu --inflight 1 --steps 20 --warmup 5 --extra-batches "" > $O/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b32.csv \;
cd $R && timeout 600 tools/pmc_run.sh 32 > $O/pmc_run.log 2>&1
python tools/pmc_summary.py $O/pmc_conv_b32.json > $O/pmc_summary.log 2>&1
tools/ubench/dma_issue > $O/ubench_dma_issue.txt 2>&1
tail -c 700 $O/bench_default.json; echo; head -8 $O/rocprof_kernel_stats_b32.csv; tail -3 $O/pmc_summary.log
