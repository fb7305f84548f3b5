#!/bin/bash
# Round evidence on the GPU box.  Everything that feeds `roofline` is taken from processes that run ONLY batch-N steps, one batch at
# a time on one stream (tools/steps_only.py), once per LAUNCH PLAN: --conc 0 = the one-batch-at-a-time plan (roofline.frac), --conc 1 =
# the plan bench.py's timed region runs with batches in flight (roofline.in_flight).  Per plan: rocprofv3 kernel stats + per-launch
# trace (-> profiles/rNN_rocprof_b32[_conc1]_summary.json, rNN_trace_launches_b32_conc{0,1}.json) and the PMC passes (HBM traffic,
# instruction mix; launch -> layer mapping from tf2_net_describe_launches).  Then the bench lines.  Outputs under gpurun_out/evidence/
# (copied to profiles/r04_* by hand).  Usage: round_evidence.sh [quick]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in "b32 32 0" "b32_conc1 32 1" "b1 1 0"; do
  set -- $V; N=$1; B=$2; CONC=$3
  rm -rf /tmp/prof_stats_$N
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$N -o ks -- python $R/tools/steps_only.py --batch $B --conc $CONC --steps 40 --meta $O/steps_$N.json > $O/rocprof_stats_$N.log 2>&1
  find /tmp/prof_stats_$N -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_$N.csv \;
  find /tmp/prof_stats_$N -name "*kernel_trace.csv" -exec cp {} /tmp/kt_$N.csv \;
  python $R/tools/rocprof_summary.py $O/rocprof_kernel_stats_$N.csv $O/steps_$N.json $O/rocprof_${N}_summary.json
  python $R/tools/trace_layers.py /tmp/kt_$N.csv $O/steps_$N.json $O/trace_launches_$N.json > $O/trace_launches_$N.txt 2>&1; tail -1 $O/trace_launches_$N.txt
done
cd $R
for CONC in 0 1; do
  timeout 600 tools/pmc_run.sh 32 $CONC pmc$CONC > $O/pmc_run$CONC.log 2>&1
  PMC_DIR=pmc$CONC PMC_CONC=$CONC python tools/pmc_summary.py $O/pmc_conv_b32_conc$CONC.json > $O/pmc_summary$CONC.log 2>&1; tail -2 $O/pmc_summary$CONC.log
done
[ "$1" = "quick" ] && exit 0
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" > $O/bench_s100.log 2>&1; tail -1 $O/bench_s100.log > $O/bench_s100.json
timeout 300 python bench.py --no-cpu --batch 64 --extra-batches "" > $O/bench_b64.log 2>&1; tail -1 $O/bench_b64.log > $O/bench_b64.json
timeout 300 python bench.py --no-cpu --batch 1 --extra-batches "" > $O/bench_b1.log 2>&1; tail -1 $O/bench_b1.log > $O/bench_b1.json
for NET in squeezenet vgg16 ssd300; do
  timeout 600 python bench.py --net $NET --steps 20 --warmup 5 --extra-batches "" --cpu-seconds 6 > $O/bench_$NET.log 2>&1; tail -1 $O/bench_$NET.log > $O/bench_$NET.json
done
timeout 200 python tools/layer_times.py --batch 32 --stamps > $O/layer_times_b32.txt 2>&1
tail -c 900 $O/bench_default.json; echo; head -8 $O/rocprof_kernel_stats_b32.csv
