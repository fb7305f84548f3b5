cd $GRAFT_REPO_ROOT
export TF2_AMD_EXP=8
for mp in 196 48 0; do for fl in 3 4 6; do
TF2_AMD_TM128_MINPIX=$mp timeout 300 python bench.py --no-cpu --steps 60 --extra-batches "" --inflight $fl 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MINPIX=$mp inflight=$fl', d['value'], d['images_per_s_one_batch_at_a_time'], d['pipeline_evidence']['latency_over_interval'])"
done; done
