#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ctx_fold or folded or gemm_wreg" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_blocks_gpu.py -x -q -k "1280" 2>&1 | tail -3
CP_DUMP=gpurun_out/cp_fold.json python tools/critical_path.py > gpurun_out/cp_fold.txt 2>&1
bash tools/ab_bench.sh gpurun_out/ab20 "BC_NO_CTX_FOLD=1" "BC_X=0"
