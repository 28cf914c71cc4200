#!/bin/bash
# scratch job: round-6 final evidence: profiles (4 configurations), parity margins (full GPU suite, -s), default bench twice, critical path
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/profile_round.sh "" > gpurun_out/prof_b1.log 2>&1
bash tools/profile_round.sh c3 --batch 8 > gpurun_out/prof_c3.log 2>&1
bash tools/profile_round.sh c5 --res 768 --batch 4 > gpurun_out/prof_c5.log 2>&1
bash tools/profile_round.sh b2 --batch 2 > gpurun_out/prof_b2.log 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu -s ) > gpurun_out/r6_parity_margins.txt 2>&1
tail -3 gpurun_out/r6_parity_margins.txt
timeout 900 python bench.py > gpurun_out/bench_final_1.json 2> gpurun_out/bench_final_1.err
timeout 900 python bench.py > gpurun_out/bench_final_2.json 2> gpurun_out/bench_final_2.err
timeout 300 python tools/critical_path.py > gpurun_out/r6_critical_path.txt 2>&1
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f'gpurun_out/bench_final_{i}.json').read().strip().splitlines()[-1])
    print(i, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('box_calibration'), {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('configs',{}).items()} )
PY
