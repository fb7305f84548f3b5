cd $GRAFT_REPO_ROOT
for cfg in "0 3" "8 4" "0 3" "8 4" "6 4" "16 4" "8 3" "5 4"; do set -- $cfg; q=$1; f=$2
if [ $q != 0 ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
timeout 200 python bench.py --no-cpu --steps 90 --inflight $f --extra-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HWQ=$q inflight=$f', d['value'], d['pipeline_evidence']['steps_resident_when_one_completes'] if d.get('pipeline_evidence') else None)"
done
