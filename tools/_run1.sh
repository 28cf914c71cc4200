#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_halo_gpu.py -x -q > gpurun_out/r5_t2.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r5_t2.log
PROBE_SHAPES=0,8,11 PROBE_ONLY=auto BC_WREG_STAMPS=1 PROBE_COLD=1 python tools/conv_probe.py 2>&1 | grep "stamps" > gpurun_out/r5_conv_stamps3.txt
grep "M=16384.*3-tile" gpurun_out/r5_conv_stamps3.txt | tail -2 | cut -c1-300
grep "M=16384.*staging" gpurun_out/r5_conv_stamps3.txt | tail -2 | cut -c1-300
grep "M=1024.*3-tile" gpurun_out/r5_conv_stamps3.txt | tail -1 | cut -c1-300
grep "M=256.*3-tile" gpurun_out/r5_conv_stamps3.txt | tail -1 | cut -c1-300
python tools/conv_repeat.py > gpurun_out/r5_conv_repeat.txt 2>&1; grep -c "0 of" gpurun_out/r5_conv_repeat.txt; grep -v "0 of" gpurun_out/r5_conv_repeat.txt | head
