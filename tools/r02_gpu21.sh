cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for n in 0 1; do
if [ $n = 1 ]; then export TF2_AMD_NO4BIT=1; fi
timeout 300 python bench.py --no-cpu --steps 20 --mode 1 --extra-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode1 NO4BIT=$n', d['value'], d['images_per_s_one_batch_at_a_time'])"
timeout 300 python bench.py --no-cpu --steps 10 --mode 2 --extra-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode2 NO4BIT=$n', d['value'], d['images_per_s_one_batch_at_a_time'])"
done
python - <<'PY'
import numpy as np
from tf2_amd import config as cfg, network, synth
t=cfg.resnet50_tables(); q=np.loadtxt('tests/golden/resnet50_Q',dtype=np.int32); m=synth.synth_model(t,q,0)
for mode in (0,1,2):
    net=network.NetWork(t); net.Quantization(synth.q_text(q)); net.LoadModel(m); net.Pack(mode)
    print('mode',mode,'packed MB',round(len(net.packed_host())/1e6,1))
PY
