#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['denoise_step_ms'], {k:v for k,v in d['configs'].items() if not isinstance(v,(dict,str))})"
python bench.py --no-cpu-baseline --no-e2e --no-configs --scheduler unipc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unipc headline', d['config']['denoise_step_ms'])"
BC_NO_CTX_FOLD=1 python bench.py --no-cpu-baseline --no-e2e --no-configs --scheduler unipc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unipc headline nofold', d['config']['denoise_step_ms'])"
