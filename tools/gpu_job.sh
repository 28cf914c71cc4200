#!/bin/bash
# scratch job (rewritten per gpurun call while developing): what the driver runs at round end - GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/t_all.log 2>&1
tail -6 gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
tail -c 600 gpurun_out/bench_driver.json
