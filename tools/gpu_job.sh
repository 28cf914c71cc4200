#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
bash tools/sweep_batch.sh gpurun_out/sw_b8 "--batch 8" "BC_X=0" "BC_HALO_CTAS=128"
bash tools/sweep_batch.sh gpurun_out/sw_768 "--res 768 --batch 4" "BC_X=0" "BC_HALO_CTAS=128"
bash tools/sweep_batch.sh gpurun_out/sw_b2b "--batch 2" "BC_X=0" "BC_HALO_CTAS=128"
