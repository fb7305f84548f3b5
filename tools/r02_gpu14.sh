cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lt
for b in 32 64 128; do timeout 300 python tools/layer_times.py --batch $b --steps 10 --out gpurun_out/lt/b$b.json > gpurun_out/lt/b$b.txt 2>&1; tail -1 gpurun_out/lt/b$b.txt; done
