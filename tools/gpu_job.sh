#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_fullsize_loop_gpu.py tests/test_configs_gpu.py -x -q -s -k "c3 or mixed or request or per_request or batch" 2>&1 | grep -E "passed|failed|C3|c3_r|Error|error" | tail -20
python - <<'PY'
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
import bench
from blobctrl_amd.pipeline import BlobCtrlEngine
dev=torch.device("cuda:0")
usd,bsd=bench.synth_weights(); ucfg,bcfg=bench.full_configs()
pipe=BlobCtrlEngine(usd,bsd,ucfg,bcfg,device="cuda:0",scheduler="ddim")
rb=bench.make_request_batch(list(range(8)),64,64,dev)
f=lambda: pipe(rb["prompt"], rb["fg"], rb["bg"], rb["score"], rb["dino"], num_inference_steps=20, guidance_scale=7.5, latents=rb["latents"], blobnet_conditioning_scale=rb["strength"])
f(); torch.cuda.synchronize()
for _ in range(2):
    t0=time.perf_counter(); f(); torch.cuda.synchronize(); print("c3 mixed ms/step", (time.perf_counter()-t0)/20*1e3)
P=pipe.plan_for(8,64,64,77,768,20,per_request=True)
print("flops/step TF", P.step_active.flops/1e12, {k:v for k,v in P.step_active.kinds.items() if k in ("conv_in","collapse")}, P.prologue.kinds.get("collapse"))
PY
