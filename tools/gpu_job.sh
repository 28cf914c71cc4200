#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
echo "--- batch 2"; bash tools/sweep_batch.sh gpurun_out/sb_a80_b2 "--batch 2" "BC_ATTN_80_4=1" "BC_X=0"
echo "--- batch 8"; bash tools/sweep_batch.sh gpurun_out/sb_a80_b8 "--batch 8" "BC_ATTN_80_4=1" "BC_X=0"
echo "--- 768 batch 4"; bash tools/sweep_batch.sh gpurun_out/sb_a80_c5 "--batch 4 --res 768" "BC_ATTN_80_4=1" "BC_X=0"
