#!/bin/bash
# Scratch driver for one gpurun call (edited per call): full GPU suite, round profile, default bench, step diagnostics.
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r5_gputest.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r5_gputest.log | head -1
bash tools/profile_round.sh > gpurun_out/r5_profile_round.log 2>&1; tail -1 gpurun_out/r5_profile_round.log | cut -c1-80
python tools/summarize_profile.py --publish gpurun_out/prof r5 | tail -1 | cut -c1-80
( time python bench.py --table ) > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; grep real gpurun_out/r5_bench_final.err
CP_DUMP=gpurun_out/r5_cp.json python tools/critical_path.py > gpurun_out/r5_critical_path.txt 2>&1
CT_DUMP=gpurun_out/r5_ct.json python tools/concurrent_timeline.py > gpurun_out/r5_concurrent_timeline.txt 2>&1
ABLATE_DUMP=gpurun_out/r5_ablate.json python tools/ablate_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_ablate.txt
head -4 gpurun_out/r5_critical_path.txt | tail -3
