#!/bin/bash
# scratch job (rewritten per call): what the driver runs at round end - GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/t_all.log 2>&1
tail -6 gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_driver.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
PY
