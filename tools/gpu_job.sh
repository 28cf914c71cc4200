#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
bash tools/gemm256_pmc.sh 2>&1 | tail -70
