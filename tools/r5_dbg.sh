#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbg; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q -k "merged_expand or squeezenet" > $O/t1.log 2>&1; grep -E "passed|failed|FAILED|layer [0-9]+|network input|Mismatched|Error" $O/t1.log | head -30
for M in 1 0 2 1 0 2; do
  TF2_AMD_TEST=1 TF2_AMD_OPTS="fire=$M" timeout 200 python bench.py --net squeezenet --no-cpu --steps 40 --warmup 5 --extra-batches "" > $O/fi_$M.log 2>&1
  python - <<EOF
import json
try:
    d=json.loads(open("$O/fi_$M.log").read().strip().splitlines()[-1]); r=d["roofline"]
    print("fire=$M", d["value"], "cold", d["cold_start"]["value"], "one-batch", d["images_per_s_one_batch_at_a_time"], "launches", r["launches_per_step"], "kus", r["kernel_us_per_step"])
except Exception as e: print("failed", e)
EOF
done
