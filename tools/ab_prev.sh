#!/bin/bash
# A/B of the previous commit (worktree _prev) against the working tree, interleaved on one box
OUT=gpurun_out/r3g; mkdir -p $OUT
for rnd in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then d=_prev; else d=.; fi
    (cd $d && timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-configs) > $OUT/${v}_r${rnd}.json 2> $OUT/${v}_r${rnd}.err
    python3 - "$OUT/${v}_r${rnd}.json" "$v" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(f"[{sys.argv[2]}] edits/s {d['value']:.4f}  step ms {d['config']['denoise_step_ms']:.3f}")
except Exception as ex:
    print(f"[{sys.argv[2]}] FAILED {ex}")
PY
  done
done
