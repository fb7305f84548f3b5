#!/bin/bash
# round 5, evidence call: the whole -m gpu suite, rocprofv3 + PMC evidence for all four networks, bench lines, DMA stress (both builds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 2400 tools/round_evidence.sh "resnet50 vgg16 ssd300 squeezenet" 1 > $O/evidence.log 2>&1; tail -25 $O/evidence.log
timeout 600 python tools/dma_stress.py --iters 200 --opts "bband_rows=4,bband_min=1;bband=2,bband_rows_alone=2,bband_min=1,alt_conc=0" --out $O/dma_stress.txt > $O/dma_stress.log 2>&1; tail -2 $O/dma_stress.txt
TF2_AMD_TOOL_LIB=1 TF2_AMD_LIB=$R/tf2_amd/libtf2amd_check.so timeout 600 python tools/dma_stress.py --iters 60 --opts "bband_rows=4,bband_min=1;bband=2,bband_rows_alone=2,bband_min=1,alt_conc=0" --out $O/dma_stress_check.txt > $O/dma_stress_check.log 2>&1; tail -3 $O/dma_stress_check.txt
