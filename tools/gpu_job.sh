#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh > gpurun_out/r5_profile_round.log 2>&1; tail -1 gpurun_out/r5_profile_round.log
python tools/summarize_profile.py --publish gpurun_out/prof r5 | tail -1
( time python bench.py --table ) > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; grep real gpurun_out/r5_bench_final.err
CP_DUMP=gpurun_out/r5_cp.json python tools/critical_path.py > gpurun_out/r5_critical_path.txt 2>&1
CT_DUMP=gpurun_out/r5_ct.json python tools/concurrent_timeline.py > gpurun_out/r5_concurrent_timeline.txt 2>&1
ABLATE_DUMP=gpurun_out/r5_ablate.json python tools/ablate_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_ablate.txt
cp profiles/r5_* gpurun_out/prof/ 2>/dev/null
head -3 gpurun_out/r5_critical_path.txt | tail -2
