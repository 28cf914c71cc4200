#!/bin/bash
# scratch job: default bench line twice on the final build
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_final_1.json 2> gpurun_out/bench_final_1.err
timeout 900 python bench.py > gpurun_out/bench_final_2.json 2> gpurun_out/bench_final_2.err
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f'gpurun_out/bench_final_{i}.json').read().strip().splitlines()[-1])
    c=d['configs']
    print(i, round(d['value'],4), round(d['ms_per_step']/50,3), d['config'].get('denoise_step_ms_normalised'), d['roofline']['frac'], d['roofline']['full_grid_launches']['frac'], d['roofline'].get('traffic'), d['roofline'].get('mfma_busy'), d['roofline'].get('avg_launch_us'), d['box_calibration'], c['unipc_ms_per_step'], c['c3_batch8_mixed_ms_per_step'], c['c3_frac_of_peak'], c['c5_768_batch4_ms_per_step'], c['c5_frac_of_peak'], c['script_default']['denoise_step_ms'], c['script_default']['edit_ms_end_to_end'], c['script_default']['loop_frac_of_peak'], d.get('edit_ms_end_to_end'), d['cpu_baseline']['value'])
PY
