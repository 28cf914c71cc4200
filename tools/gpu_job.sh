#!/bin/bash
# scratch job: row-chain staging with all row chunks requested first: tests + A/B batch 1 / 8
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_rowchain_gpu.py tests/test_blocks_gpu.py -x -q > gpurun_out/t_k.log 2>&1
tail -3 gpurun_out/t_k.log
for rep in 1 2 3; do
for lib in "" "$GRAFT_REPO_ROOT/build/ab/lib_prev.so"; do
  BLOBCTRL_HIP_LIB="$lib" timeout 600 python bench.py --steps 3 --warmup 2 --no-calibration --no-cpu-baseline --no-e2e --no-configs > gpurun_out/b1.json 2> gpurun_out/b1.err || tail -5 gpurun_out/b1.err
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print('batch1 lib[%s] ms/step %.3f'%(sys.argv[1][-12:], d['ms_per_step']/50))
PY
done; done
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/build/ab/lib_prev.so"; do
  BLOBCTRL_HIP_LIB="$lib" timeout 600 python bench.py --batch 8 --steps 6 --warmup 2 --no-calibration > gpurun_out/b8.json 2> gpurun_out/b8.err || tail -5 gpurun_out/b8.err
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b8.json').read().strip().splitlines()[-1]); print('batch8 lib[%s] ms/step %.2f'%(sys.argv[1][-12:], d['ms_per_step']/50))
PY
done; done
