#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
ABLATE_DUMP=gpurun_out/r5_ablate.json python tools/ablate_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_ablate.txt; cat gpurun_out/r5_ablate.txt
