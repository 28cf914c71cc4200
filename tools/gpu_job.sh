#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/soak.py 2>&1 | tail -1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -1 gpurun_out/profile_round.log
