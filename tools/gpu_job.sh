#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python tools/critical_path.py > gpurun_out/critical_path.txt 2>&1
python tools/concurrent_timeline.py > gpurun_out/concurrent_timeline.txt 2>&1
tail -3 gpurun_out/concurrent_timeline.txt
