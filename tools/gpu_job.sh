#!/bin/bash
# scratch job: BlobNet's M = 512 ff.net.0 on gemm256 or not (batch 1) + calibration probes with the latency chain
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
for plan in "" "g256_min_tiles=128"; do
  BC_PLAN="$plan" timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-configs > gpurun_out/b1.json 2> gpurun_out/b1.err || tail -5 gpurun_out/b1.err
  python - "$plan" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print('batch1 plan[%s] ms/step %.3f'%(sys.argv[1], d['ms_per_step']/50), d['box_calibration'])
PY
done; done
