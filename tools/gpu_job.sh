#!/bin/bash
# scratch job: round-6 final evidence on the final sources: warm-up, profiles (4 configurations), parity margins (full GPU suite, -s), critical path, default bench twice
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python bench.py --steps 2 --warmup 1 --no-calibration --no-cpu-baseline --no-e2e --no-configs > /dev/null 2>&1
bash tools/profile_round.sh "" > gpurun_out/prof_b1.log 2>&1
bash tools/profile_round.sh c3 --batch 8 > gpurun_out/prof_c3.log 2>&1
bash tools/profile_round.sh c5 --res 768 --batch 4 > gpurun_out/prof_c5.log 2>&1
bash tools/profile_round.sh b2 --batch 2 > gpurun_out/prof_b2.log 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu -s ) > gpurun_out/r6_parity_margins.txt 2>&1
tail -3 gpurun_out/r6_parity_margins.txt
timeout 300 python tools/critical_path.py > gpurun_out/r6_critical_path.txt 2>&1
