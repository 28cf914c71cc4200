#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/critical_path.py > gpurun_out/r6_critical_path.txt 2>&1
tail -45 gpurun_out/r6_critical_path.txt
