cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['by_path'], d['roofline']['frac'])"
TF2_AMD_NOFUSE=1 timeout 300 python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NOFUSE', d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['by_path'], d['roofline']['frac'])"
