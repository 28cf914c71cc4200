#!/bin/bash
# scratch job: 4-stage K / V^T pipeline for the one-workgroup-per-CU D = 80 attention: tests, the launch alone, A/B in the step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or attention" > gpurun_out/t_k.log 2>&1
tail -3 gpurun_out/t_k.log
python tools/attn_probe.py 2>&1 | grep "d= 80\|lib"
python tools/attn_probe.py build/ab/lib_prev.so 2>&1 | grep "d= 80\|lib"
for rep in 1 2 3; do
for lib in "" "$GRAFT_REPO_ROOT/build/ab/lib_prev.so"; do
  BLOBCTRL_HIP_LIB="$lib" timeout 600 python bench.py --steps 3 --warmup 2 --no-calibration --no-cpu-baseline --no-e2e --no-configs > gpurun_out/b1.json 2> gpurun_out/b1.err || tail -5 gpurun_out/b1.err
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print('batch1 lib[%s] ms/step %.3f'%(sys.argv[1][-12:], d['ms_per_step']/50))
PY
done; done
