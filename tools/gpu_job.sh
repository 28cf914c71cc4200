#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rowchain_gpu.py tests/test_blocks_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -4
BC_RC_STAMPS=1 PROBE_COLD=1 python tools/rowchain_probe.py 2>&1 | grep "stamps out\|BlobNet\|UNet" | tail -4 | cut -c1-330
bash tools/ab_bench.sh gpurun_out/ab23 "BLOBCTRL_HIP_LIB=$GRAFT_REPO_ROOT/_prev/blobctrl_amd/libblobctrl_hip.so" "BC_X=0"
