#!/bin/bash
# Round 2, second GPU session: fused bottleneck kernel -- parity first, then per-layer times and bench lines.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or unfused or every_layer_batch2 or batch32_logits" > $O/pytest_fused.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fused.log; tail -15 $O/pytest_fused.log
timeout 200 python tools/layer_times.py --batch 32 > $O/lt_fused.txt 2>&1; tail -60 $O/lt_fused.txt
TF2_AMD_FUSE_SHAPE=0 timeout 200 python tools/layer_times.py --batch 32 > $O/lt_fused_s0.txt 2>&1; grep -E "^( 3|13|16|26|32|35) " $O/lt_fused_s0.txt
TF2_AMD_FUSE_SHAPE=1 timeout 200 python tools/layer_times.py --batch 32 > $O/lt_fused_s1.txt 2>&1; grep -E "^( 3|13|16|26|32|35) " $O/lt_fused_s1.txt
TF2_AMD_NOFUSE=1 timeout 200 python tools/layer_times.py --batch 32 > $O/lt_nofuse.txt 2>&1; tail -1 $O/lt_nofuse.txt
timeout 300 python bench.py --no-cpu --steps 60 > $O/bench_fused.log 2>&1; tail -1 $O/bench_fused.log > $O/bench_fused.json
python -c "
import json; d=json.load(open('$O/bench_fused.json')); print({k:d[k] for k in ('value','ms_per_step','images_per_s_one_batch_at_a_time','images_per_s_by_batch','latency_batch1')}); print(d['roofline']['achieved'], d['roofline']['frac'])"
