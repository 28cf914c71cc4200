#!/bin/bash
# Ablation bits of conv_halo.hip (wrong results by design): needs a library built with -DBC_DIAGNOSTICS
#   HIPCC_EXTRA=-DBC_DIAGNOSTICS python __graft_entry__.py --force    (round 6: not reachable from a production build)
for d in 0 1 2 3 4 8 12 16 7 15; do echo "DBG=$d"; BC_PLAN=wreg=0 BC_HALO_DBG=$d PROBE_SHAPES=0,4,8,11 PROBE_ONLY=1 timeout 120 python tools/conv_probe.py 2>&1 | grep "^B"; done
