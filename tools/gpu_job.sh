#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "gemm256" -x -q 2>&1 | tail -8
bash tools/sweep_batch.sh gpurun_out/sw8 "--batch 8" "BC_PLAN=g256=0" "BC_X=1" 2>&1 | tail -4
bash tools/sweep_batch.sh gpurun_out/sw5 "--res 768 --batch 4" "BC_PLAN=g256=0" "BC_X=1" 2>&1 | tail -4
