#!/bin/bash
# Round evidence on the GPU box.  Everything that feeds `roofline` is taken from processes that run ONLY batch-N steps one batch at
# a time (tools/steps_only.py): rocprofv3 kernel stats (-> profiles/rNN_rocprof_b32_summary.json, what bench.py reports as
# roofline.kernel_us_per_step_rocprof), the same at batch 1, PMC passes for HBM traffic and instruction mix (launch -> layer
# mapping from tf2_net_describe_launches).  Then the bench lines and the per-layer HIP-event table.  Outputs under
# gpurun_out/evidence/ (copied to profiles/r03_* by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 32 1; do
  rm -rf /tmp/prof_stats_$B
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$B -o ks -- python $R/tools/steps_only.py --batch $B --steps 40 --meta $O/steps_b$B.json > $O/rocprof_stats_b$B.log 2>&1
  find /tmp/prof_stats_$B -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b$B.csv \;
  python $R/tools/rocprof_summary.py $O/rocprof_kernel_stats_b$B.csv $O/steps_b$B.json $O/rocprof_b${B}_summary.json
done
cd $R && timeout 600 tools/pmc_run.sh 32 > $O/pmc_run.log 2>&1
python tools/pmc_summary.py $O/pmc_conv_b32.json > $O/pmc_summary.log 2>&1; tail -2 $O/pmc_summary.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" > $O/bench_s100.log 2>&1; tail -1 $O/bench_s100.log > $O/bench_s100.json
timeout 300 python bench.py --no-cpu --batch 64 --extra-batches "" > $O/bench_b64.log 2>&1; tail -1 $O/bench_b64.log > $O/bench_b64.json
timeout 300 python bench.py --no-cpu --batch 1 --extra-batches "" > $O/bench_b1.log 2>&1; tail -1 $O/bench_b1.log > $O/bench_b1.json
timeout 200 python tools/layer_times.py --batch 32 --stamps > $O/layer_times_b32.txt 2>&1
timeout 200 python tools/layer_times.py --batch 1 > $O/layer_times_b1.txt 2>&1
tail -c 900 $O/bench_default.json; echo; head -8 $O/rocprof_kernel_stats_b32.csv
