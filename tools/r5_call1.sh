#!/bin/bash
# round 5, GPU call 1: the whole -m gpu suite, the DMA stress (product build, then the TF2_CHECK_DMA build), a bench line, the other networks' evidence
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python tools/dma_stress.py --iters 200 --opts "bband_rows=4,bband_min=1;bband=2,bband_rows_alone=2,bband_min=1,alt_conc=0" --out $O/dma_stress.txt > $O/dma_stress.log 2>&1; tail -2 $O/dma_stress.txt
TF2_AMD_TOOL_LIB=1 TF2_AMD_LIB=$R/tf2_amd/libtf2amd_check.so timeout 600 python tools/dma_stress.py --iters 60 --opts "bband_rows=4,bband_min=1;bband=2,bband_rows_alone=2,bband_min=1,alt_conc=0" --out $O/dma_stress_check.txt > $O/dma_stress_check.log 2>&1; tail -3 $O/dma_stress_check.txt
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 6 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json; tail -c 400 $O/bench_default.json; echo
timeout 1500 tools/round_evidence.sh "ssd300 squeezenet vgg16" 0 > $O/evidence.log 2>&1; tail -12 $O/evidence.log
