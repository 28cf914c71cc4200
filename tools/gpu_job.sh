#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
timeout 900 python tools/soak.py 2>&1 | tail -4
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log
