#!/bin/bash
# scratch job: pipelined attention (QK of tile t+1 ahead of tile t's softmax) for the D = 80 small-grid launches: correctness + the launch alone
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python - <<'PY'
import sys, os
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import attn_probe
from blobctrl_amd import _lib
lib = _lib.load()
for B in (1, 2):
    attn_probe.run(lib, B, 8, 80, 2048, 2048, check=True)
    attn_probe.run(lib, B, 8, 80, 1024, 1024, check=True)
    attn_probe.run(lib, B, 8, 80, 256, 2048 + 64, check=True)
PY
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or attention" > gpurun_out/t_k.log 2>&1
tail -3 gpurun_out/t_k.log
