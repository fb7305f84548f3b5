#!/bin/bash
# Round evidence on the GPU box: bench line (with cpu_baseline), rocprofv3 kernel stats of the same command,
# PMC passes for HBM traffic.  The kernel stats are taken one batch at a time (--inflight 1): that is the condition
# under which bench.py measures the per-kernel durations behind `roofline` (overlapping batches stretch them).  Outputs under gpurun_out/evidence/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
python bench.py --no-cpu --batch 64 --streams 2 --extra-batches "" > $O/bench_b64_s2.log 2>&1
python bench.py --no-cpu --batch 1 --extra-batches "" > $O/bench_b1.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o ks -- python $R/bench.py --no-cpu --inflight 1 --steps 20 --warmup 5 --extra-batches "" > $O/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b32.csv \;
find /tmp/prof_stats -name "*domain_stats.csv" -exec cp {} $O/rocprof_domain_stats_b32.csv \;
cd $R && tools/pmc_run.sh 32 > $O/pmc_run.log 2>&1
python tools/pmc_summary.py $O/pmc_conv_b32.json > $O/pmc_summary.log 2>&1
tail -2 $O/bench_default.log; tail -1 $O/bench_b64_s2.log; tail -1 $O/bench_b1.log; head -12 $O/rocprof_kernel_stats_b32.csv; tail -3 $O/pmc_summary.log
