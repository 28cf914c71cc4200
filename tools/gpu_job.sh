#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
BC_RC_PAIR=7 timeout 900 python -m pytest tests/test_rowchain_gpu.py tests/test_blocks_gpu.py -x -q 2>&1 | tail -5
bash tools/ab_bench.sh gpurun_out/ab22 "BC_X=0" "BC_RC_PAIR=2" "BC_RC_PAIR=3"
