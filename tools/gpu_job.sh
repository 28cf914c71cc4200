#!/bin/bash
# scratch job: round-6 final evidence: profiles (4 configurations), parity margins (full GPU suite, -s), default bench twice, critical path, probes
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
timeout 300 python tools/blas_ceiling.py > gpurun_out/r6_blas_ceiling.txt 2>&1
timeout 300 python tools/g256_probe.py > gpurun_out/r6_g256_probe.txt 2>&1
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f'gpurun_out/bench_final_{i}.json').read().strip().splitlines()[-1])
    c=d['configs']
    print(i, round(d['value'],4), round(d['ms_per_step']/50,3), d['config'].get('denoise_step_ms_normalised'), d['roofline']['frac'], d['roofline']['full_grid_launches']['frac'], d['roofline'].get('traffic'), d['box_calibration'], c['unipc_ms_per_step'], c['c3_batch8_mixed_ms_per_step'], c['c3_frac_of_peak'], c['c5_768_batch4_ms_per_step'], c['c5_frac_of_peak'], c['script_default']['denoise_step_ms'], c['script_default']['edit_ms_end_to_end'], c['script_default']['loop_frac_of_peak'], d.get('edit_ms_end_to_end'), d['cpu_baseline']['value'])
PY
