cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for m in 200 100000 200 100000; do
TF2_AMD_ALT_MIN=$m timeout 300 python bench.py --no-cpu --steps 90 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ALT_MIN=$m', d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['by_path'], d['roofline']['frac'], d['roofline']['in_flight']['frac'])"
done
