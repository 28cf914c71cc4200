#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_blocks_gpu.py -q -x 2>&1 | tail -6
bash tools/ab_bench.sh gpurun_out/ab11 "BC_NO_FF2_PROJ_OUT=1" "BC_X=0"
