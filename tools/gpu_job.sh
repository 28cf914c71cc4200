#!/bin/bash
# scratch job: GroupNorm pass form as default for >= 4 requests: block tests + full-size batch tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_blocks_gpu.py tests/test_fullsize_loop_gpu.py tests/test_dist_gpu.py -x -q -s > gpurun_out/t_gn.log 2>&1
tail -5 gpurun_out/t_gn.log; grep -n "C3 request\|c5\|768" gpurun_out/t_gn.log | head
