#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 3; do echo "NBAND_MB=$v"; BC_X_NBAND_MB=$v PROBE_ONLY=auto PROBE_SHAPES=15,16 PROBE_COLD=1 timeout 300 python tools/conv_probe.py 2>&1 | grep "^B"; done
bash tools/sweep_batch.sh gpurun_out/sw8 "--batch 8" "BC_X=1" "BC_X_NBAND_MB=3" 2>&1 | tail -4
