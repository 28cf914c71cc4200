"""Oracle (TEST INFRASTRUCTURE): the denoise loop of StableDiffusionBlobNetPipeline.__call__ restated.

Restates blobctrl/pipelines/pipeline_blobnet.py:724-739 (construct_blobnet_input: channel-cat then WIDTH-cat,
left = clean reference latents, right = noisy latents), :1006-1012 (blobnet_keep), :1025-1102 (loop body:
BlobNet -> right-square slices -> UNet -> crop right half -> CFG -> scheduler.step).  VAE / CLIP / image
pre-processing are outside the hot path: this takes image latents, prompt embeddings and the
DINOv2 vector as inputs.
"""
import torch

from .nets import NetConfig, blobnet_forward, unet_forward


def construct_input(latent_model_input, scores, image_latents, feats=None):
    """pipe:724-739."""
    if feats is not None:
        right = torch.cat([latent_model_input, scores, feats], dim=1)
        left = torch.cat([image_latents, scores, feats], dim=1)
    else:
        right = torch.cat([latent_model_input, scores], dim=1)
        left = torch.cat([image_latents, scores], dim=1)
    return torch.cat([left, right], dim=-1)


def blobnet_keep(num_steps, start, end):
    """pipe:1006-1012."""
    return [1.0 - float(i / num_steps < start or (i + 1) / num_steps > end) for i in range(num_steps)]


def noise_pred_step(unet_sd, unet_cfg, blob_sd, blob_cfg, latents, t, prompt_embeds, fg_lat, bg_lat,
                    fg_score, bg_score, fg_feats, cond_scale, guidance_scale):
    """One loop body up to (and including) CFG: pipe:1031-1098.  All image-side tensors are already repeated to 2B."""
    lmi = torch.cat([latents] * 2)                                           # pipe:1031 (scale_model_input = identity)
    blob_in = construct_input(lmi, fg_score, fg_lat, fg_feats)
    down, mid, up = blobnet_forward(blob_sd, blob_cfg, blob_in, t, cond_scale)
    unet_in = construct_input(lmi, bg_score, bg_lat)
    sq = lambda r: r[..., -r.shape[-2]:]                                     # pipe:1085-1087
    eps = unet_forward(unet_sd, unet_cfg, unet_in, t, prompt_embeds,
                       [sq(r) for r in down], sq(mid), [sq(r) for r in up])
    h, w = eps.shape[-2:]
    eps = eps[..., :h, w // 2:]                                              # pipe:1092-1093
    eu, ec = eps.chunk(2)
    return eu + guidance_scale * (ec - eu)                                   # pipe:1096-1098


def denoise_loop(unet_sd, unet_cfg: NetConfig, blob_sd, blob_cfg: NetConfig, scheduler, num_steps, latents,
                 prompt_embeds, fg_lat, bg_lat, gs_score, dino_vec, guidance_scale=7.5,
                 conditioning_scale=1.0, guidance_start=0.0, guidance_end=1.0, trace=None):
    """pipe:953-1102 minus VAE/CLIP.  latents [B,4,h,w]; prompt_embeds [2B,77,D] = cat(neg, pos);
    fg_lat / bg_lat [1,4,h,w] (already x0.18215); gs_score [1,2,h,w]; dino_vec [1,1,Cf]."""
    B2 = prompt_embeds.shape[0]
    scheduler.set_timesteps(num_steps)
    latents = latents * scheduler.init_noise_sigma
    fg_lat = fg_lat.repeat(B2, 1, 1, 1)
    bg_lat = bg_lat.repeat(B2, 1, 1, 1)
    bg_score, fg_score = gs_score.unbind(dim=1)                              # pipe:974
    bg_score = bg_score.unsqueeze(1).repeat(B2, 1, 1, 1).to(latents.dtype)
    fg_score = fg_score.unsqueeze(1).repeat(B2, 1, 1, 1).to(latents.dtype)
    feats = dino_vec.repeat(B2, 1, 1).to(latents.dtype)
    fg_feats = torch.einsum("nmhw,nmc->nchw", fg_score, feats).contiguous()  # pipe:984
    keep = blobnet_keep(num_steps, guidance_start, guidance_end)
    for i, t in enumerate(scheduler.timesteps):
        cond = conditioning_scale * keep[i]
        eps = noise_pred_step(unet_sd, unet_cfg, blob_sd, blob_cfg, latents, t, prompt_embeds, fg_lat, bg_lat,
                              fg_score, bg_score, fg_feats, cond, guidance_scale)
        if trace is not None:
            trace.append((latents.clone(), eps.clone()))
        latents = scheduler.step(eps, latents)
    return latents
