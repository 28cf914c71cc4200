#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh > gpurun_out/r5_profile_round.log 2>&1; tail -5 gpurun_out/r5_profile_round.log
ls gpurun_out/prof | head
CP_DUMP=gpurun_out/r5_cp.json python tools/critical_path.py > gpurun_out/r5_critical_path.txt 2>&1; head -4 gpurun_out/r5_critical_path.txt
CT_DUMP=gpurun_out/r5_ct.json python tools/concurrent_timeline.py > gpurun_out/r5_concurrent_timeline.txt 2>&1; head -3 gpurun_out/r5_concurrent_timeline.txt
