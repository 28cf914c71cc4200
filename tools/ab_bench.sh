#!/bin/bash
# A/B of whole-step variants on ONE box, interleaved: usage  bash tools/ab_bench.sh OUTDIR "ENV_A" "ENV_B" ...
OUT=$1; shift
mkdir -p $OUT
for rnd in 1 2; do
  i=0
  for e in "$@"; do
    env $e timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-configs > $OUT/v${i}_r${rnd}.json 2> $OUT/v${i}_r${rnd}.err
    python3 - "$OUT/v${i}_r${rnd}.json" "$e" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(f"[{sys.argv[2]}] edits/s {d['value']:.4f}  step ms {d['config']['denoise_step_ms']:.3f}")
except Exception as ex:
    print(f"[{sys.argv[2]}] FAILED {ex}")
PY
    i=$((i+1))
  done
done
