#!/bin/bash
# Scratch driver for one gpurun call (edited per call).
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rowchain_gpu.py tests/test_blocks_gpu.py -x -q 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, torch, bench
from blobctrl_amd import synth
from blobctrl_amd.vae import AutoencoderKL
from blobctrl_amd.pipeline import BlobCtrlEngine
dev = torch.device("cuda:0")
ucfg, bcfg = bench.full_configs()
usd, bsd = bench.synth_weights()
vae = AutoencoderKL(synth.synth_state_dict(synth.vae_param_shapes(), 33), device=str(dev))
eng = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="unipc", vae=vae)
for B in (1, 2):
    lat = torch.randn(B, 4, 64, 64, device=dev)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        img = eng.decode_latents(lat, "pt")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        z = vae.decode(lat / 0.18215, return_dict=False)[0]
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"B={B} rep {rep}: decode_latents {1e3*(t1-t0):.2f} ms, vae.decode {1e3*(t2-t1):.2f} ms")
PY
