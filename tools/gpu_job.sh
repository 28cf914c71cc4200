#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
bash tools/ab_prev.sh
