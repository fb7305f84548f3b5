cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "resnet50 or tiny or googlenet or squeezenet or pruned" > gpurun_out/r3e/test1.log 2>&1; tail -5 gpurun_out/r3e/test1.log
for u in 1 0; do
timeout 200 env TF2_AMD_AVG_FUSE=$u python bench.py --steps 20 --warmup 5 --no-cpu --extra-batches "" > gpurun_out/r3e/bench_u$u.log 2>&1; tail -1 gpurun_out/r3e/bench_u$u.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('avg_fuse $u', d['value'], d['images_per_s_one_batch_at_a_time'], d['roofline']['frac'], d['latency_batch1']['us_per_image'])"
done
