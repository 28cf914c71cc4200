#!/bin/bash
# (GPU) compile and run tools/attn_bound.hip: the issue-bound ceiling of the D = 40 attention tile with all operands in registers.
# Writes gpurun_out/attn_bound.json; tools/summarize_attn_bound.py turns it into the MFMA-busy ceiling per waves / SIMD.
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/attn_bound.hip -o gpurun_out/attn_bound 2> gpurun_out/attn_bound.build.log || { cat gpurun_out/attn_bound.build.log; exit 1; }
./gpurun_out/attn_bound > gpurun_out/attn_bound.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/attn_bound.json"))
runs = d["runs"]
base = {r["waves_per_simd"]: r for r in runs if r["softmax_valu"] == 0}
print(f"device {d['device']}, {d['cus']} CUs; matrix-pipe cycles per 32q x 64k tile: {d['mfma_cycles_per_tile']}")
best0 = min(r["ns_per_tile_per_simd"] for r in base.values())          # MFMA-only, pipe saturated: the 100 % reference at the held clock
for r in runs:
    if r["softmax_valu"]:
        w = r["waves_per_simd"]
        print(f"  {w} wave(s) per SIMD: {r['ns_per_tile_per_simd']:.1f} ns per tile with the softmax VALU work, {base[w]['ns_per_tile_per_simd']:.1f} ns MFMAs only "
              f"-> MFMA-busy ceiling {100 * best0 / r['ns_per_tile_per_simd']:.0f} % ({r['mfma_tflops_padded']:.0f} TFLOP/s of padded MFMA work)")
PY
rm -f gpurun_out/attn_bound
