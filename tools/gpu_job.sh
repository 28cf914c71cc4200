#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python bench.py --batch 8 --steps 1 --warmup 1 --denoise-steps 10 --no-cpu-baseline --no-e2e --no-configs --table > gpurun_out/t8.json 2> gpurun_out/t8.err
python - <<'PY'
import json
s=open('gpurun_out/t8.err').read()
i=s.index('{\n "by_kernel"')
d,_=json.JSONDecoder().raw_decode(s[i:])
for k,v in d['by_kernel'].items():
    if v['ms']>0.1: print(f"{k[:60]:60s} n={v['launches']:4d} ms={v['ms']:.3f} TF={v['tflops']}")
for r in d['top_shapes']:
    if 'gemm' in r['variant']:
        print(f"{r['kind']:10s} {r['variant'][:44]:44s} {str(r['shape']):40s} n={r['launches']:3d} us={1e3*r['ms']/r['launches']:.1f} TF={r['tflops']}")
PY
