#!/bin/bash
# scratch job: batch-8 A/B: row-chain vs unfused blocks on gemm256
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for plan in "" "rowchain=0" "" "rowchain=0"; do
  BC_PLAN="$plan" timeout 600 python bench.py --batch 8 --steps 10 --warmup 3 --no-calibration > gpurun_out/b8.json 2> gpurun_out/b8.err || tail -5 gpurun_out/b8.err
  python - "$plan" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b8.json').read().strip().splitlines()[-1]); print('batch8 plan[%s] ms/step %.2f'%(sys.argv[1], d['ms_per_step']/50), d['config'].get('plan'))
PY
done
