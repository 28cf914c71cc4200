#!/bin/bash
# scratch job: fused GroupNorm prologue vs GroupNorm pass + plain conv_wreg at batch 8 / 1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PROBE_ONLY=auto PROBE_SPLIT=1 PROBE_SHAPES=14,15,16,0,4 timeout 900 python tools/conv_probe.py 2>&1 | tail -40
