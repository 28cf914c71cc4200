"""PyTorch custom-op registrations of the hot path (`torch.ops.blobctrl.*`, `torch.library.custom_op`).

BASELINE.json's north star words the boundary as "driven from Python through PyTorch-ROCm custom ops that keep the
StableDiffusionBlobCtrlPipeline.__call__ / BlobNet / UNet-LoRA API surface".  The compute entry points of this package are the C ABI
of libblobctrl_hip.so (include/blobctrl_hip.h) reached through ctypes; this module registers the four calls the reference makes into
its hot path as dispatcher-visible ops OVER that same C ABI, so that `torch.profiler`, dispatch modes and fake-tensor tracing see them:

    blobctrl::splat_scores     blobctrl/utils/utils.py:145-194             (splat_features, tuple score_size branch)
    blobctrl::blobnet_forward  blobctrl/models/blobnet.py:720-945          (BlobNetModel.forward -> down / mid / up residuals)
    blobctrl::unet_forward     D/models/unets/unet_2d_condition.py:1039-1353 (patched forward with the three residual lists)
    blobctrl::denoise          blobctrl/pipelines/pipeline_blobnet.py:1025-1123 (the whole loop: one hipGraph launch per edit)

The nn.Module shells (modules.py), `splat.splat_features` and `BlobCtrlEngine.__call__` call THROUGH these ops.  A module / engine is
passed as an integer handle (ops take tensors and scalars, not Python objects): `register(obj)` keeps a weak reference.  Every op has
a fake (meta) kernel that states its output shapes, dtypes and devices without touching the GPU; the real kernels raise
BlobCtrlHipError when the HIP library is missing - there is no CPU fallback.  No autograd: the path is inference only.
"""
import weakref
from typing import List, Optional

import torch

_objects = weakref.WeakValueDictionary()
_next_handle = [1]


def register(obj) -> int:
    """Handle of a module / engine for the ops below (kept weakly: the handle dies with the object)."""
    h = getattr(obj, "_bc_handle", None)
    if h is None or _objects.get(h) is not obj:
        h = _next_handle[0]
        _next_handle[0] += 1
        _objects[h] = obj
        try:
            object.__setattr__(obj, "_bc_handle", h)
        except Exception:          # noqa: BLE001  (objects without a __dict__)
            pass
    return h


def _get(handle: int):
    obj = _objects.get(int(handle))
    if obj is None:
        raise RuntimeError(f"blobctrl op: handle {handle} does not name a live module / engine (ops.register)")
    return obj


def _dev(index: int) -> torch.device:
    return torch.device("cuda", index)


# ------------------------------------------------------------------------------------------------------------------ splat_scores
@torch.library.custom_op("blobctrl::splat_scores", mutates_args=())
def splat_scores(params: torch.Tensor, h: int, w: int, device_index: int) -> torch.Tensor:
    """params [n][8] float64 on the host (xs, ys, cov00, cov01, cov10, cov11, size, unused) -> scores [n][2][h][w] float64 on the GPU."""
    from . import splat
    return splat._splat_scores_impl(params, h, w, _dev(device_index))


@splat_scores.register_fake
def _(params, h, w, device_index):
    return torch.empty((params.shape[0], 2, h, w), dtype=torch.float64, device=_dev(device_index))


# ------------------------------------------------------------------------------------------------------------------ blobnet_forward
def blobnet_output_shapes(cfg, B: int, H: int, W: int):
    """(channels, h, w) of BlobNet's 12 down + 1 mid + 12 up residuals on an H x W canvas (bn:860-864, 881, 921-924)."""
    boc = cfg.block_out_channels
    nb = len(boc)
    down, sizes = [(boc[0], H, W)], []
    h, w = H, W
    for i in range(nb):
        sizes.append((h, w))
        down += [(boc[i], h, w)] * cfg.layers_per_block
        if i < nb - 1:
            h, w = (h + 1) // 2, (w + 1) // 2
            down.append((boc[i], h, w))
    mid = (boc[-1], h, w)
    rev = list(reversed(boc))
    up = []
    for i in range(nb):
        hh, ww = sizes[nb - 1 - i]
        up += [(rev[i], hh, ww)] * (cfg.layers_per_block + 1)
        if i < nb - 1:
            up.append((rev[i],) + sizes[nb - 2 - i])
    return down, mid, up


@torch.library.custom_op("blobctrl::blobnet_forward", mutates_args=())
def blobnet_forward(sample: torch.Tensor, timestep: float, conditioning_scale: float, handle: int) -> List[torch.Tensor]:
    """BlobNetModel.forward: the down residuals, the mid residual, the up residuals, in that order (NCHW, sample's dtype)."""
    down, mid, up = _get(handle)._forward_impl(sample, timestep, conditioning_scale)
    return list(down) + [mid] + list(up)


@blobnet_forward.register_fake
def _(sample, timestep, conditioning_scale, handle):
    m = _get(handle)
    B, _, H, W = sample.shape
    down, mid, up = blobnet_output_shapes(m.trunk_config, B, H, W)
    dt = sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float32
    return [torch.empty((B,) + s, dtype=dt, device=m.device) for s in down + [mid] + up]


# ------------------------------------------------------------------------------------------------------------------ unet_forward
@torch.library.custom_op("blobctrl::unet_forward", mutates_args=())
def unet_forward(sample: torch.Tensor, timestep: float, encoder_hidden_states: torch.Tensor, down_add: List[torch.Tensor],
                 mid_add: Optional[torch.Tensor], up_add: List[torch.Tensor], handle: int) -> torch.Tensor:
    """The patched UNet2DConditionModel.forward; empty lists / None = no BlobNet residuals.  Returns eps [B][4][H][W] fp32."""
    with_res = mid_add is not None
    return _get(handle)._forward_impl(sample, timestep, encoder_hidden_states, list(down_add) if with_res else None, mid_add,
                                      list(up_add) if with_res else None)


@unet_forward.register_fake
def _(sample, timestep, encoder_hidden_states, down_add, mid_add, up_add, handle):
    m = _get(handle)
    B, _, H, W = sample.shape
    return torch.empty((B, m.trunk_config.out_channels, H, W), dtype=torch.float32, device=m.device)


# ------------------------------------------------------------------------------------------------------------------ denoise
@torch.library.custom_op("blobctrl::denoise", mutates_args=())
def denoise(prompt_embeds: torch.Tensor, fg_image_latents: torch.Tensor, bg_image_latents: torch.Tensor, gs_score: torch.Tensor,
            dino_feats: torch.Tensor, latents: torch.Tensor, num_inference_steps: int, guidance_scale: float,
            conditioning_scales: List[float], guidance_start: float, guidance_end: float, handle: int,
            scales_are_per_request: bool = False) -> torch.Tensor:
    """The denoise loop of one edit (or one request batch) at tensor level: final latents [B][4][h][w] fp32 on the engine's device.
    `scales_are_per_request`: the caller passed a LIST of conditioning scales (one per request; `denoise` validates its length exactly
    as for a direct call) rather than one float (a one-element list used to be collapsed to a scalar and broadcast)."""
    eng = _get(handle)
    sc = list(conditioning_scales) if scales_are_per_request else conditioning_scales[0]
    return eng.denoise(prompt_embeds, fg_image_latents, bg_image_latents, gs_score, dino_feats, num_inference_steps=num_inference_steps,
                       guidance_scale=guidance_scale, latents=latents, blobnet_conditioning_scale=sc,
                       blobnet_control_guidance_start=guidance_start, blobnet_control_guidance_end=guidance_end)


@denoise.register_fake
def _(prompt_embeds, fg_image_latents, bg_image_latents, gs_score, dino_feats, latents, num_inference_steps, guidance_scale,
      conditioning_scales, guidance_start, guidance_end, handle, scales_are_per_request=False):
    eng = _get(handle)
    return torch.empty(tuple(latents.shape), dtype=torch.float32, device=eng.device)
