#!/usr/bin/env python3
"""How much does the 50-step edit amplify a perturbation of its start latents?  (a free-running full-size
comparison against the CPU reference needs a step map that does not blow small differences up.)

For conv_out scales s (the UNet's last convolution; eps = s * eps_1) the engine runs the 512x512 edit from x0 and from
x0 + 1e-3 * N(0,1) and reports |x_final|, the relative divergence of the final latents and the amplification factor
= (divergence / scale of x_final) / (perturbation / scale of x0).  GPU box:  python tools/amplification_probe.py
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def main():
    import bench
    from blobctrl_amd import synth
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import splat_features
    dev = "cuda:0"
    h = w = 64
    ucfg, bcfg = bench.full_configs()
    inp = bench.synth_inputs(h, w)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=dev)
    usd0, bsd0 = bench.synth_weights()
    d = torch.from_numpy(__import__("numpy").random.Generator(__import__("numpy").random.PCG64(99)).standard_normal((1, 4, h, w)).astype("float32"))
    out = []
    scales = [float(s) for s in (sys.argv[1:] or ["1.0", "0.5", "0.3", "0.2", "0.1"])]
    for s in scales:
        usd, bsd = synth.contractive_variant(usd0, bsd0, conv_out_scale=s)
        for sched in ("ddim", "unipc"):
            eng = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=dev, scheduler=sched)
            run = lambda x: eng(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=50, guidance_scale=7.5,
                                latents=x, blobnet_control_guidance_end=1.0).float().cpu()
            a = run(inp["latents"])
            b = run(inp["latents"] + 1e-3 * d)
            a2 = run(inp["latents"])
            div = (a - b).abs().max().item() / a.abs().max().item()
            pert = 1e-3 * d.abs().max().item() / inp["latents"].abs().max().item()
            rec = dict(conv_out_scale=s, scheduler=sched, x_final_max=round(a.abs().max().item(), 2), x_final_std=round(a.std().item(), 3),
                       rel_divergence=div, rel_perturbation=pert, amplification=div / pert, deterministic=bool(torch.equal(a, a2)))
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del eng
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "amplification.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
