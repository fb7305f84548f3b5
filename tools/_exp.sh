#!/bin/bash
mkdir -p gpurun_out
run() { echo -n "$1: "; env $2 timeout 300 python bench.py --no-cpu --steps 5 --warmup 3 --extra-batches "" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['latency_batch1']['by_path'])"; }
run default X=1
run bneck_min1 TF2_AMD_BNECK_MIN=1
run g56f TF2_AMD_BGROUP_MIN56F=1
run g56f+bneck "TF2_AMD_BGROUP_MIN56F=1 TF2_AMD_BNECK_MIN=1"
run g56 "TF2_AMD_BGROUP_MIN56F=1 TF2_AMD_BGROUP_MIN56=1"
run g7 TF2_AMD_BGROUP_MIN7=1
run default X=1
