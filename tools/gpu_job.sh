#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh > gpurun_out/prof_b1.log 2>&1
bash tools/profile_round.sh c3 --batch 8 > gpurun_out/prof_c3.log 2>&1
bash tools/profile_round.sh c5 --res 768 --batch 4 > gpurun_out/prof_c5.log 2>&1
bash tools/profile_round.sh b2 --batch 2 > gpurun_out/prof_b2.log 2>&1
ls gpurun_out/prof gpurun_out/prof_c3
python tools/ablate_probe.py > gpurun_out/r6_ablate.log 2>&1; tail -30 gpurun_out/r6_ablate.log
