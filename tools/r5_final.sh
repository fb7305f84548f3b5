#!/bin/bash
# round 5, closing call(s).  Args: a list of stages out of: tests sq_evidence bench_all bench_sq stress
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd $R
for S in "$@"; do case $S in
tests)
  timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log;;
sq_evidence)
  timeout 900 tools/round_evidence.sh "squeezenet" 0 > $O/evidence_sq.log 2>&1; tail -4 $O/evidence_sq.log
  E=$R/gpurun_out/evidence
  for f in rocprof_squeezenet_b32_conc1_summary.json rocprof_squeezenet_b32_summary.json trace_launches_squeezenet_b32.json trace_launches_squeezenet_b32_conc1.json \
           rocprof_kernel_stats_squeezenet_b32.csv rocprof_kernel_stats_squeezenet_b32_conc1.csv pmc_conv_squeezenet_b32_conc1.json; do
    [ -s $E/$f ] && cp $E/$f $R/profiles/r05_$f
  done;;
bench_sq)
  timeout 600 python bench.py --net squeezenet --steps 20 --warmup 5 --extra-batches "" --cpu-seconds 6 > $O/bench_squeezenet.log 2>&1; tail -1 $O/bench_squeezenet.log > $O/bench_squeezenet.json; tail -c 300 $O/bench_squeezenet.json; echo;;
bench_all)
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; tail -c 300 $O/bench_default.json; echo
  timeout 300 python bench.py --no-cpu --steps 100 --warmup 10 --extra-batches "" > $O/bench_s100.log 2>&1; tail -1 $O/bench_s100.log > $O/bench_s100.json
  for NET in vgg16 ssd300; do
    timeout 600 python bench.py --net $NET --steps 20 --warmup 5 --extra-batches "" --cpu-seconds 6 > $O/bench_$NET.log 2>&1; tail -1 $O/bench_$NET.log > $O/bench_$NET.json; tail -c 200 $O/bench_$NET.json; echo
  done;;
stress)
  timeout 600 python tools/dma_stress.py --iters 100 --nets squeezenet,vgg16 --out $O/dma_stress_sq.txt > $O/dma_stress_sq.log 2>&1; tail -2 $O/dma_stress_sq.txt;;
esac; done
