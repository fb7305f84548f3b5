#!/bin/bash
# SQ counter passes for the launches of ONE kernel (name substring $1, batch $2): instruction mix, pipe activity, LDS conflicts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-conv_stem}; B=${2:-32}
rm -rf $R/gpurun_out/pmck; mkdir -p $R/gpurun_out/pmck
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmck/sq1 -- python $R/tools/layer_times.py --batch $B --steps 1 > $R/gpurun_out/pmck/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmck/sq2 -- python $R/tools/layer_times.py --batch $B --steps 1 > $R/gpurun_out/pmck/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmck/sq3 -- python $R/tools/layer_times.py --batch $B --steps 1 > $R/gpurun_out/pmck/sq3.log 2>&1
python - <<PY
import csv, glob, collections
R="$R"; K="$K"
def table(d):
    fs = glob.glob(f"{R}/gpurun_out/pmck/{d}/*/*_counter_collection.csv")
    if not fs: return []
    disp = collections.OrderedDict()
    for x in csv.DictReader(open(fs[0])):
        e = disp.setdefault(int(x["Dispatch_Id"]), {"kernel": x["Kernel_Name"], "grid": int(x["Grid_Size"])})
        e[x["Counter_Name"]] = float(x["Counter_Value"])
    return [v for v in disp.values() if K in v["kernel"]][-1:]
r = {}
for d in ("sq1", "sq2", "sq3"):
    for v in table(d): r.update(v)
print(r.pop("kernel", "?")[:90])
print({k: int(v) for k, v in r.items()})
PY
