#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_wreg" 2>&1 | tail -5
BC_GW_PREFER=64 python -m pytest tests/test_blocks_gpu.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh gpurun_out/ab16 "BC_X=0" "BC_GW_PREFER=64"
