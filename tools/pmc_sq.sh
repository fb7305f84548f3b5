#!/bin/bash
# SQ counter passes only (instruction mix / pipe activity per conv launch); the option string TF2_AMD_OPTS is inherited
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
B=${1:-32}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc/sq1 -- python $R/tools/layer_times.py --batch $B --steps 1 > $R/gpurun_out/pmc/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc/sq2 -- python $R/tools/layer_times.py --batch $B --steps 1 > $R/gpurun_out/pmc/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc/sq3 -- python $R/tools/layer_times.py --batch $B --steps 1 > $R/gpurun_out/pmc/sq3.log 2>&1
python - <<PY
import csv, glob, collections
R="$R"
def table(d):
    fs = glob.glob(f"{R}/gpurun_out/pmc/{d}/*/*_counter_collection.csv")
    if not fs: return []
    disp = collections.OrderedDict()
    for x in csv.DictReader(open(fs[0])):
        e = disp.setdefault(int(x["Dispatch_Id"]), {"kernel": x["Kernel_Name"], "grid": int(x["Grid_Size"])})
        e[x["Counter_Name"]] = float(x["Counter_Value"])
    return [v for v in disp.values() if "conv_mfma" in v["kernel"]][-54:]
a, b, c = table("sq1"), table("sq2"), table("sq3")
for i in (1, 2, 3, 4, 12, 13, 14, 26, 28):
    r = dict(a[i]); r.update(b[i] if b else {}); r.update(c[i] if c else {})
    k = r.pop("kernel").split("(")[0][-40:]
    print(i, k, {kk: int(v) for kk, v in r.items()})
PY
