cd $GRAFT_REPO_ROOT
for m in 256 500 256 500; do
TF2_AMD_BNECK_MIN=$m timeout 300 python bench.py --no-cpu --steps 60 --batch 64 --extra-batches "128" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BNECK_MIN=$m b64', d['value'], d['images_per_s_one_batch_at_a_time'], d['images_per_s_by_batch'])"
done
