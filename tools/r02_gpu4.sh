#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or every_layer_batch2 or batch32_logits" > $O/pytest_bneck.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bneck.log; tail -12 $O/pytest_bneck.log
timeout 200 python tools/layer_times.py --batch 32 > $O/lt_bneck.txt 2>&1
TF2_AMD_NOFUSE=1 timeout 200 python tools/layer_times.py --batch 32 > $O/lt_nofuse.txt 2>&1
paste <(awk '{print $1,$2,$3,$4,$5,$10}' $O/lt_nofuse.txt) <(awk '{print $10}' $O/lt_bneck.txt) | head -58
tail -1 $O/lt_nofuse.txt; tail -1 $O/lt_bneck.txt
for nf in 1 0; do if [ $nf = 1 ]; then export TF2_AMD_NOFUSE=1; else unset TF2_AMD_NOFUSE; fi; timeout 300 python bench.py --no-cpu --steps 60 2>&1 | tail -1 > $O/bench_nofuse$nf.json; python -c "
import json; d=json.load(open('$O/bench_nofuse$nf.json')); print('NOFUSE=$nf', d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['us_per_image'])"; done
