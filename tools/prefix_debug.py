"""(GPU, scratch) where does the CFG-prefix plan differ from the plain plan?  eps of ONE step per image and per canvas half."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from blobctrl_amd.pipeline import BlobCtrlEngine
from blobctrl_amd.splat import splat_features

usd, bsd = bench.synth_weights()
ucfg, bcfg = bench.full_configs()
h = w = 64
inp = bench.synth_inputs(h, w, batch=1)
score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device="cuda:0")
eps = []
for plain in (False, True):
    if plain:
        os.environ["BC_NO_CFG_PREFIX"] = "1"
    pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device="cuda:0", scheduler="ddim")
    pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=1, guidance_scale=7.5, latents=inp["latents"])
    P = pipe.plan_for(1, h, w, 77, 768, 1)
    torch.cuda.synchronize()
    eps.append(P.eps_active.clone().float().cpu().view(2, 64, 128, 4))
    print("kinds", {k: v for k, v in P.step_active.kinds.items() if k in ("dup_halves", "rowchain", "attention")})
a, b = eps
for img in range(2):
    d = (a[img] - b[img]).abs()
    print(f"image {img}: max diff {d.max():.4e} (|eps| max {b[img].abs().max():.3f}); left half {d[:, :64].max():.3e} right half {d[:, 64:].max():.3e}; "
          f"mean diff {d.mean():.3e}")
