#!/bin/bash
# scratch job: ff.net.0 on gemm256 by default: full GPU suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -4 gpurun_out/t_all.log
