#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
timeout 40 python -m pytest tests/test_gpu_configs.py -q -x -k "graph_replay" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_stats_32
timeout 45 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_32 -o ks -- python $R/tools/steps_only.py --batch 32 --steps 40 --spinup-ms 300 --meta $O/steps_b32.json > $O/rocprof_stats_b32.log 2>&1
find /tmp/prof_stats_32 -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_b32.csv \;
python $R/tools/rocprof_summary.py $O/rocprof_kernel_stats_b32.csv $O/steps_b32.json $O/rocprof_b32_summary.json
