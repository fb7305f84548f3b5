#!/bin/bash
# prefetching K loop: parity (every ResNet50 layer + batch 32 logits + VGG + SSD small), then A/B per-layer times and bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_layer or batch32 or wave_shapes or squeezenet or vgg" > $O/pytest_pf.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pf.log; tail -6 $O/pytest_pf.log
timeout 200 python tools/layer_times.py --batch 32 > $O/lt_pf.txt 2>&1
TF2_AMD_EXP=8 timeout 200 python tools/layer_times.py --batch 32 > $O/lt_nopf.txt 2>&1
TF2_AMD_EXP=16 timeout 200 python tools/layer_times.py --batch 32 > $O/lt_pfall.txt 2>&1
paste <(awk '{print $1,$2,$3,$4,$5,$8,$10}' $O/lt_nopf.txt) <(awk '{print $10}' $O/lt_pf.txt) <(awk '{print $10}' $O/lt_pfall.txt) | head -60
tail -1 $O/lt_nopf.txt; tail -1 $O/lt_pf.txt; tail -1 $O/lt_pfall.txt
for e in 8 0; do TF2_AMD_EXP=$e timeout 300 python bench.py --no-cpu --steps 60 --extra-batches "" > $O/bench_exp$e.log 2>&1; tail -1 $O/bench_exp$e.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('EXP=$e', d['value'], d['images_per_s_one_batch_at_a_time'], d['latency_batch1']['us_per_image'], d['pipeline_evidence'])"; done
