#!/bin/bash
# Round evidence on the GPU box.  Everything that feeds `roofline` is taken from processes that run ONLY batch-N steps, one batch at
# a time on one stream (tools/steps_only.py), once per LAUNCH PLAN: --conc 1 = the plan bench.py's timed region runs with batches in
# flight (top level of `roofline`), --conc 0 = the one-batch-at-a-time plan (`roofline.one_batch`).  Per plan: rocprofv3 kernel stats +
# per-launch trace (-> profiles/r06_rocprof_[<net>_]b32[_conc1]_summary.json, r06_trace_launches_[<net>_]b32[_conc1].json) and the PMC
# passes (HBM traffic, instruction mix; launch -> row mapping from tf2_net_describe_launches -> r06_pmc_conv_[<net>_]b32[_conc1].json).
# Round 5: the same for the other BASELINE.json networks (vgg16, ssd300, squeezenet); round 6: and for the reference's other two shipped networks (googlenet, resnet50_pruned).  Outputs under gpurun_out/evidence/ (copied to
# profiles/r06_* by hand).  Usage: round_evidence.sh [nets, default "resnet50 vgg16 ssd300 squeezenet"] [bench: 1 = also the bench lines]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
NETS=${1:-"resnet50 vgg16 ssd300 squeezenet googlenet resnet50_pruned"}; BENCH=${2:-1}
cd /tmp && export TMPDIR=/tmp
for NET in $NETS; do
  T=""; [ $NET != resnet50 ] && T="${NET}_"
  PLANS="b32_conc1:32:1 b32:32:0"; [ $NET = resnet50 ] && PLANS="$PLANS b1:1:0"
  for V in $PLANS; do
    IFS=: read N B CONC <<< "$V"; N="${T}${N}"
    rm -rf /tmp/prof_stats_$N
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$N -o ks -- python $R/tools/steps_only.py --net $NET --batch $B --conc $CONC --steps 40 --meta $O/steps_$N.json > $O/rocprof_stats_$N.log 2>&1
    find /tmp/prof_stats_$N -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_$N.csv \;
    find /tmp/prof_stats_$N -name "*kernel_trace.csv" -exec cp {} /tmp/kt_$N.csv \;
    python $R/tools/rocprof_summary.py $O/rocprof_kernel_stats_$N.csv $O/steps_$N.json $O/rocprof_${N}_summary.json
    python $R/tools/trace_layers.py /tmp/kt_$N.csv $O/steps_$N.json $O/trace_launches_$N.json > $O/trace_launches_$N.txt 2>&1; tail -1 $O/trace_launches_$N.txt
  done
  cd $R
  PC="1 0"; [ $NET != resnet50 ] && PC="1"       # (the other networks: the PMC passes of the in-flight plan only)
  for CONC in $PC; do
    S=""; [ $CONC = 1 ] && S="_conc1"
    timeout 600 tools/pmc_run.sh 32 $CONC pmc_${NET}$CONC $NET > $O/pmc_run_${NET}$CONC.log 2>&1
    PMC_NET=$NET PMC_DIR=pmc_${NET}$CONC PMC_CONC=$CONC python tools/pmc_summary.py $O/pmc_conv_${T}b32$S.json > $O/pmc_summary_${NET}$CONC.log 2>&1; tail -1 $O/pmc_summary_${NET}$CONC.log
  done
  cd /tmp
done
# the summaries bench.py attaches to its lines (rocprof_summary / traffic), into profiles/ of THIS copy of the tree before the bench legs
for f in $O/rocprof_*_summary.json $O/trace_launches_*.json $O/pmc_conv_*.json $O/rocprof_kernel_stats_*.csv; do
  [ -s $f ] && cp $f $R/profiles/r06_$(basename $f)
done
[ "$BENCH" = "1" ] || exit 0
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" > $O/bench_s100.log 2>&1; tail -1 $O/bench_s100.log > $O/bench_s100.json
for NET in squeezenet vgg16 ssd300 googlenet resnet50_pruned; do
  timeout 600 python bench.py --net $NET --steps 20 --warmup 5 --extra-batches "" --cpu-seconds 6 > $O/bench_$NET.log 2>&1; tail -1 $O/bench_$NET.log > $O/bench_$NET.json
done
timeout 600 python bench.py --mode 1 --steps 10 --warmup 3 --cpu-seconds 4 --extra-batches "" > $O/bench_mode1.log 2>&1; tail -1 $O/bench_mode1.log > $O/bench_mode1.json
tail -c 600 $O/bench_default.json; echo
