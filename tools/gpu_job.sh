#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh gpurun_out/ab21 "BC_X=0" "BC_SPLIT_CFG=1" "BC_NO_GW=1" "BC_NO_ROWCHAIN=1"
timeout 600 python tools/conv_repeat.py 2>&1 | tail -3
