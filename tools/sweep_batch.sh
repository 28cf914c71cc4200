#!/bin/bash
# One-box sweep of planning constants at another batch / resolution: usage  bash tools/sweep_batch.sh OUTDIR "BENCH ARGS" "ENV_A" "ENV_B" ...
# (baseline = first entry; two interleaved rounds; 10 denoise steps per edit)
OUT=$1; shift
ARGS=$1; shift
mkdir -p $OUT
for rnd in 1 2; do
  i=0
  for e in "$@"; do
    env $e timeout 600 python bench.py $ARGS --steps 1 --warmup 1 --denoise-steps 10 --no-cpu-baseline --no-roofline --no-e2e --no-configs > $OUT/v${i}_r${rnd}.json 2> $OUT/v${i}_r${rnd}.err
    python3 - "$OUT/v${i}_r${rnd}.json" "$e" "$rnd" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(f"r{sys.argv[3]} [{sys.argv[2]}] step ms {d['config']['denoise_step_ms']:.3f}", flush=True)
except Exception as ex:
    print(f"r{sys.argv[3]} [{sys.argv[2]}] FAILED {ex}", flush=True)
PY
    i=$((i+1))
  done
done
