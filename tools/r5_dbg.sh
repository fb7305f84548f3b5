#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbg; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q -k "first_layer_with or first_3x3 or tiny" > $O/t1.log 2>&1; grep -E "passed|failed|FAILED|layer [0-9]+|network input|Mismatched" $O/t1.log | head -30
timeout 1500 tools/round_evidence.sh "vgg16 ssd300" 0 > $O/evidence.log 2>&1; tail -8 $O/evidence.log
for NET in vgg16 ssd300; do timeout 600 python bench.py --net $NET --steps 20 --warmup 5 --extra-batches "" --cpu-seconds 6 > $O/bench_$NET.log 2>&1; tail -1 $O/bench_$NET.log > $R/gpurun_out/evidence/bench_$NET.json; tail -c 300 $R/gpurun_out/evidence/bench_$NET.json; echo; done
