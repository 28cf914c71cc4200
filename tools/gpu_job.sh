#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gemm8p_probe.sh > /dev/null 2>&1; cat gpurun_out/gemm8p_probe.txt | tail -14
python tools/blas_ceiling.py 2>&1 | tee gpurun_out/r6_blas_ceiling.txt | tail -12
bash tools/profile_round.sh > gpurun_out/prof_b1.log 2>&1
bash tools/profile_round.sh c3 --batch 8 > gpurun_out/prof_c3.log 2>&1
bash tools/profile_round.sh c5 --res 768 --batch 4 > gpurun_out/prof_c5.log 2>&1
bash tools/profile_round.sh b2 --batch 2 > gpurun_out/prof_b2.log 2>&1
python tools/critical_path.py > gpurun_out/r6_critical_path.txt 2>&1
python tools/concurrent_timeline.py > gpurun_out/r6_concurrent_timeline.txt 2>&1; tail -5 gpurun_out/r6_concurrent_timeline.txt
