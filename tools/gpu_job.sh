#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -2 gpurun_out/bench_final.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_final.json"))
print(d["value"], d["config"]["denoise_step_ms"], d["config"].get("denoise_step_ms_normalised"), d.get("box_calibration"))
print(d["roofline"]["frac"], d["roofline"]["full_grid_launches"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"], d["roofline"]["mfma_busy"], d["roofline"]["step_traffic_gb"], d["roofline"]["step_algorithmic_gb"], d["config"]["plan"]["launches_per_active_step"], d["config"]["algorithmic_tflop_per_edit"])
print({k:v for k,v in d["configs"].items() if not isinstance(v,(dict,str))})
print(d["configs"]["script_default"]["edit_ms_end_to_end"], d["configs"]["script_default"]["denoise_step_ms"], d["configs"]["script_default"]["loop_frac_of_peak"], d["configs"]["script_default"]["images_per_s"], d["end_to_end"]["edit_ms_end_to_end"], d["cpu_baseline"]["s_per_step"], d["cpu_baseline"].get("threads"), d.get("errors"))
PY
