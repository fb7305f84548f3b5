cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resnet50" 2>&1 | tail -2
timeout 300 python tools/layer_times.py --batch 32 2>&1 | grep -E "^ 0 |total" 
timeout 300 python tools/layer_times.py --batch 1 2>&1 | grep -E "^ 0 |total" 
bash tools/pmc_kernel.sh conv_stem 32
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/pmck/sq1/runc/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_stem' in r['Kernel_Name']:
            print('stem us', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
PY
