#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbg; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "whole_window or vgg16 or first_3x3 or vgg_small or squeezenet" > $O/t1.log 2>&1; grep -E "passed|failed|FAILED|layer [0-9]+|network input" $O/t1.log | head -30
cd /tmp && export TMPDIR=/tmp
for NET in squeezenet vgg16; do
  rm -rf /tmp/ps_$NET; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$NET -o ks -- python $R/tools/steps_only.py --net $NET --batch 32 --conc 1 --steps 30 --meta $O/steps_$NET.json > $O/rp_$NET.log 2>&1
  find /tmp/ps_$NET -name "*kernel_stats.csv" -exec cp {} $O/ks_$NET.csv \;
  python - <<EOF
import csv
rows=list(csv.DictReader(open("$O/ks_$NET.csv")))
for r in rows[:12]:
    if "tf2::" in r["Name"]: print("$NET", r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,2))
EOF
done
