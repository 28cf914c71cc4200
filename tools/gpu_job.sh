#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_final.json"))
print(d["value"], d["config"]["denoise_step_ms"], d["roofline"]["frac"], d["roofline"]["full_grid_launches"], d["roofline"]["traffic"], d["roofline"]["mfma_busy"], d["roofline"]["avg_launch_us"])
print({k:v for k,v in d["configs"].items() if not isinstance(v,(dict,str))})
print(d["configs"]["script_default"]["edit_ms_end_to_end"], d["configs"]["script_default"]["denoise_step_ms"], d["end_to_end"]["edit_ms_end_to_end"], d["cpu_baseline"]["s_per_step"])
PY
