#!/usr/bin/env python3
"""End-to-end edit on the HIP path, the stages of scripts/blobctrl_inference.py in order (inf:112-205):
   ellipse -> splat scores | fg crop -> DINOv2 pooled feature | prompt ids -> CLIP embeddings | fg/bg images -> VAE latents
   -> 50-step blob-conditioned denoise loop -> VAE decode -> image.
With --models DIR it loads the released layout (README.md:115-128: DIR/stable-diffusion-v1-5/{unet,vae,text_encoder},
DIR/blobnet, DIR/unet_lora, DIR/dinov2-large); without it, seeded synthetic weights of the same architectures are used (no
checkpoints exist offline), which exercises every kernel and prints per-stage timings but produces noise, not a picture."""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default=None)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--out", default=None, help="write the edited image as a .npy [H,W,3] float array")
    args = ap.parse_args()
    import bench
    from blobctrl_amd import synth
    from blobctrl_amd.clip_text import CLIPTextModel
    from blobctrl_amd.dinov2 import Dinov2Model
    from blobctrl_amd.modules import BlobNetModel, UNet2DConditionModel
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import blob_dict_from_ellipse, splat_features
    from blobctrl_amd.vae import AutoencoderKL

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    t0 = sync()
    if args.models:
        sd15 = os.path.join(args.models, "stable-diffusion-v1-5")
        unet = UNet2DConditionModel.from_pretrained(sd15, subfolder="unet", extra_in_channels=1, lora_path=os.path.join(args.models, "unet_lora"))
        blob = BlobNetModel.from_pretrained(os.path.join(args.models, "blobnet"))
        vae = AutoencoderKL.from_pretrained(sd15, subfolder="vae")
        clip = CLIPTextModel.from_pretrained(sd15, subfolder="text_encoder")
        dino = Dinov2Model.from_pretrained(os.path.join(args.models, "dinov2-large"))
        pipe = BlobCtrlEngine(unet.weights, blob.weights, unet.trunk_config, blob.trunk_config, scheduler="unipc", vae=vae,
                                              text_encoder=clip)
    else:
        ucfg, bcfg = bench.full_configs()
        usd, bsd = bench.synth_weights()
        vae = AutoencoderKL(synth.synth_state_dict(synth.vae_param_shapes(), 33))
        clip = CLIPTextModel(synth.synth_state_dict(synth.clip_text_param_shapes(), 88), num_heads=12)
        dino = Dinov2Model(synth.synth_state_dict(synth.dinov2_param_shapes(1024, 24, 4, 14, 37 * 37), 99), num_heads=16)
        pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, scheduler="unipc", vae=vae, text_encoder=clip)
    t1 = sync()
    R = args.res
    rng = np.random.Generator(np.random.PCG64(0))
    fg = torch.from_numpy(rng.uniform(-1, 1, (1, 3, R, R)).astype(np.float32))
    bg = torch.from_numpy(rng.uniform(-1, 1, (1, 3, R, R)).astype(np.float32))
    s = R / 512.0
    ellipse = [[361.1067 * s, 367.8526 * s], [85.4812 * s, 103.6543 * s], 87.3739]            # the README's move_hat target blob
    score = splat_features(**blob_dict_from_ellipse(ellipse, R, R), score_size=(R // 8, R // 8), return_d_score=True)
    t2 = sync()
    crop = torch.nn.functional.interpolate(fg, size=(224, 224), mode="bilinear", align_corners=False)
    feats = dino(crop).pooler_output.view(1, 1, -1)
    t3 = sync()
    ids = torch.from_numpy(rng.integers(0, 49408, size=(2, 1, 77)).astype(np.int64))
    prompt = pipe.encode_prompt(ids[1], ids[0])
    t4 = sync()
    img = pipe(prompt, None, None, score, feats, num_inference_steps=args.steps, guidance_scale=7.5,
               generator=torch.Generator().manual_seed(0), blobnet_conditioning_scale=1.0, blobnet_control_guidance_end=0.9,
               fg_image=fg, bg_image=bg, output_type="np")
    t5 = sync()
    print(f"models {t1 - t0:.1f} s | splat {1e3 * (t2 - t1):.1f} ms | DINOv2 {1e3 * (t3 - t2):.1f} ms | CLIP {1e3 * (t4 - t3):.1f} ms | "
          f"VAE encode x2 + {args.steps}-step loop + VAE decode {1e3 * (t5 - t4):.1f} ms (first call: includes planning + graph capture)")
    t6 = sync()
    img = pipe(prompt, None, None, score, feats, num_inference_steps=args.steps, guidance_scale=7.5,
               generator=torch.Generator().manual_seed(0), blobnet_conditioning_scale=1.0, blobnet_control_guidance_end=0.9,
               fg_image=fg, bg_image=bg, output_type="np")
    t7 = sync()
    print(f"second edit (plans cached): {1e3 * (t7 - t6):.1f} ms end to end; image {img.shape}, range [{img.min():.3f}, {img.max():.3f}]")
    if args.out:
        np.save(args.out, img[0])


if __name__ == "__main__":
    main()
