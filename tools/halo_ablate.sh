for d in 0 1 2 3 4 8 12 16 7 15; do echo "DBG=$d"; BC_HALO_DBG=$d PROBE_SHAPES=0,4,8,11 PROBE_ONLY=1 timeout 120 python tools/conv_probe.py 2>&1 | grep "^B"; done
