#!/bin/bash
# scratch job: round-6 profiles on the final kernel sources (batch 1 / 8 / 768^2 x 4 / batch 2)
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh "" > gpurun_out/prof_b1.log 2>&1; tail -2 gpurun_out/prof_b1.log
bash tools/profile_round.sh c3 --batch 8 > gpurun_out/prof_c3.log 2>&1; tail -2 gpurun_out/prof_c3.log
bash tools/profile_round.sh c5 --res 768 --batch 4 > gpurun_out/prof_c5.log 2>&1; tail -2 gpurun_out/prof_c5.log
bash tools/profile_round.sh b2 --batch 2 > gpurun_out/prof_b2.log 2>&1; tail -2 gpurun_out/prof_b2.log
