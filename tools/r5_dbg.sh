#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbg; mkdir -p $O; cd $R
TF2_AMD_TEST=1 TF2_AMD_OPTS="c3_tiles32=1" timeout 300 python -m pytest tests -m gpu -q -k "vgg16_full or vgg16_batch32" > $O/t1.log 2>&1; grep -E "passed|failed|FAILED|layer [0-9]+|Mismatched" $O/t1.log | head
for NET in vgg16 ssd300; do for M in 1 0 1 0; do
  TF2_AMD_TEST=1 TF2_AMD_OPTS="c3_tiles32=$M" timeout 200 python bench.py --net $NET --no-cpu --steps 40 --warmup 5 --extra-batches "" > $O/ct_${NET}_$M.log 2>&1
  python - <<EOF
import json
try:
    d=json.loads(open("$O/ct_${NET}_$M.log").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$NET c3_tiles32=$M", d["value"], "cold", d["cold_start"]["value"], "one-batch", d["images_per_s_one_batch_at_a_time"], "kus", r["kernel_us_per_step"])
except Exception as e: print("failed", e)
EOF
done; done
