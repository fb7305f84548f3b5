cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu --steps 90 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['by_path'], d['roofline']['frac'], d['roofline']['in_flight']['frac'])"; }
run base
TF2_AMD_SK=2 run nosk
TF2_AMD_SK8=0 run sk4only
TF2_AMD_SK8=100000 run sk8always
run base
