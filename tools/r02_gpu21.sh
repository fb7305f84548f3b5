cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiny or mode1 or squeezenet or vgg_small" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu --steps 20 --mode 1 --extra-batches "" 2>&1 | tail -1 | tee gpurun_out/bench_mode1.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode1', d['value'], d['images_per_s_one_batch_at_a_time'])"
timeout 300 python bench.py --no-cpu --steps 10 --mode 2 --extra-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode2', d['value'], d['images_per_s_one_batch_at_a_time'])"
