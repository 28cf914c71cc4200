#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_halo_gpu.py -q 2>&1 | tail -2
