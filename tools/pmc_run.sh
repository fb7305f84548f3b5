#!/bin/bash
# PMC passes for the conv kernels (run on the GPU box): per-dispatch counters as CSV under gpurun_out/pmc/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-32}; CONC=${2:-0}; D=${3:-pmc}; NET=${4:-resnet50}      # batch, launch plan (0 one batch at a time / 1 the batches-in-flight plan), output dir under gpurun_out/, network
rm -rf $R/gpurun_out/$D; mkdir -p $R/gpurun_out/$D
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/$D/sq1 -- python $R/tools/steps_only.py --spinup-ms 0 --net $NET --batch $B --conc $CONC --steps 2 > $R/gpurun_out/$D/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/$D/sq2 -- python $R/tools/steps_only.py --spinup-ms 0 --net $NET --batch $B --conc $CONC --steps 2 > $R/gpurun_out/$D/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$D/tcc1 -- python $R/tools/steps_only.py --spinup-ms 0 --net $NET --batch $B --conc $CONC --steps 2 > $R/gpurun_out/$D/tcc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$D/tcc2 -- python $R/tools/steps_only.py --spinup-ms 0 --net $NET --batch $B --conc $CONC --steps 2 > $R/gpurun_out/$D/tcc2.log 2>&1
find $R/gpurun_out/$D -name "*.csv" | head -20
