#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_halo_gpu.py tests/test_vae_gpu.py -x -q 2>&1 | tail -2
bash tools/ab_bench.sh gpurun_out/ab9 "BC_WREG_NT=0" "BC_X=0" "BC_WREG_NT=1"
python tools/conv_repeat.py > gpurun_out/r5_conv_repeat.txt 2>&1; grep -c "0 of" gpurun_out/r5_conv_repeat.txt; grep -v "0 of\|amdgpu" gpurun_out/r5_conv_repeat.txt | head -3
