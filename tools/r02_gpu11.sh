cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/layer_times.py --batch 32 2>&1 | grep -E "total" 
timeout 300 python bench.py --no-cpu --steps 90 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['by_path'], d['roofline']['frac'])"
