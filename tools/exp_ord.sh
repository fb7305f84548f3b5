cd $GRAFT_REPO_ROOT
for e in 0 32 64; do TF2_AMD_EXP=$e timeout 200 python tools/layer_times.py --batch 32 > /tmp/lt_$e.txt 2>&1; done
paste <(awk '{print $1,$2,$3,$4,$5,$8,$10}' /tmp/lt_0.txt) <(awk '{print $10}' /tmp/lt_32.txt) <(awk '{print $10}' /tmp/lt_64.txt) | head -58
tail -1 /tmp/lt_0.txt; tail -1 /tmp/lt_32.txt; tail -1 /tmp/lt_64.txt
TF2_AMD_EXP=32 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_layer_batch2 or batch32" 2>&1 | tail -2
for e in 0 32 64; do TF2_AMD_EXP=$e timeout 300 python bench.py --no-cpu --steps 60 --extra-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('EXP=$e', d['value'], d['images_per_s_one_batch_at_a_time'])"; done
