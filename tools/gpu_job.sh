#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_halo_gpu.py -x -q 2>&1 | tail -4
bash tools/ab_bench.sh gpurun_out/ab6 "BC_X=0" "BC_WREG4=1" "BC_WREG4=2"
