cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for e in 1 0; do
TF2_AMD_STEM=$e timeout 300 python tools/layer_times.py --batch 32 2>&1 | grep -E "^ 0 |^ 1 |total" 
TF2_AMD_STEM=$e timeout 300 python tools/layer_times.py --batch 1 2>&1 | grep -E "^ 0 |total" 
TF2_AMD_STEM=$e timeout 300 python bench.py --no-cpu --steps 60 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('STEM=$e', d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'], d['latency_batch1']['by_path'], d['roofline']['frac'])"
done
