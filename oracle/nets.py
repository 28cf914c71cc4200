"""Oracle (TEST INFRASTRUCTURE): functional PyTorch-fp32 restatement of the patched SD-1.5 UNet and BlobNet.

Operates on plain state-dicts using the reference's parameter names, so the same
weights feed the real reference classes (in `tools/make_golden.py`), this oracle and the HIP engine.

Restated reference code (paths relative to /root/reference, D/ = diffusers/src/diffusers/):
  * D/models/embeddings.py:27-78, 576-588            sinusoidal embedding + TimestepEmbedding MLP
  * D/models/resnet.py:320-373                        ResnetBlock2D.forward
  * D/models/transformers/transformer_2d.py:479-527   continuous-input Transformer2DModel
  * D/models/attention.py:421-541, 1161-1167          BasicTransformerBlock / FeedForward
  * D/models/activations.py:113-123                   GEGLU (value = first half, gate = second half)
  * D/models/attention_processor.py:2154-2236         AttnProcessor2_0 (SDPA, scale d^-1/2)
  * D/models/upsampling.py:141-184, downsampling.py:132-149
  * D/models/unets/unet_2d_blocks.py:1241-1323, 1378-1433, 860-899, 2514-2624, 2677-2765 (patched blocks)
  * D/models/unets/unet_2d_condition.py:1039-1353     patched UNet2DConditionModel.forward
  * blobctrl/models/blobnet.py:720-945                BlobNetModel.forward
"""
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

# the oracle's attention is the call the reference makes (F.scaled_dot_product_attention); the explicit
# softmax(QK^T)V form materialised a [2, 8, 8192, 8192] fp32 score tensor per L0 self-attention (4.3 GB) and made the CPU baseline
# 3x slower than the reference it stands for.  BC_ORACLE_EXPLICIT_ATTENTION=1 keeps the explicit form for cross-checks.
EXPLICIT_ATTENTION = bool(os.environ.get("BC_ORACLE_EXPLICIT_ATTENTION"))


@dataclass
class NetConfig:
    """Subset of the diffusers config that the hot path depends on (bn:152-205, unet_2d_condition.py:171-230)."""
    in_channels: int = 5                   # UNet: 4 latent + 1 score;  BlobNet: 4 + conditioning_channels
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_heads: int = 8                     # diffusers `attention_head_dim=8` is really the head COUNT for SD-1.5
    norm_num_groups: int = 32
    cross_attention_dim: Optional[int] = 768   # None => BlobNet (attn2/norm2 absent)
    # down: CrossAttn x3 + Down; up: Up + CrossAttn x3 (fixed SD-1.5 topology, bn:159-172)

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


# ----------------------------------------------------------------------------------------------- leaf ops

def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """embeddings.py:27-78 with flip_sin_to_cos=True, downscale_freq_shift=0 (bn:795, unet :909-933)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - 0.0)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def time_embed(sd, timestep, batch, cfg: NetConfig):
    t = torch.as_tensor(timestep).reshape(-1).expand(batch)
    t_emb = timestep_embedding(t, cfg.block_out_channels[0])
    h = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    h = F.silu(h)
    return F.linear(h, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def resnet_block(sd, p, x, temb, groups, eps=1e-5):
    """resnet.py:320-373 (time_embedding_norm='default', output_scale_factor=1)."""
    h = F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    t = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def attention(sd, p, x, ctx, heads):
    """attention_processor.py:2154-2236: to_q/k/v (no bias), SDPA, to_out.0 (+bias)."""
    B, N, C = x.shape
    src = x if ctx is None else ctx
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(src, sd[p + "to_k.weight"])
    v = F.linear(src, sd[p + "to_v.weight"])
    d = C // heads
    q = q.view(B, -1, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    if EXPLICIT_ATTENTION:                                   # the same arithmetic with the [B, heads, N, N] score tensor materialised
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), v)
    else:                                                    # what the reference calls (attention_processor.py:2216-2218)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, C)
    return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def basic_transformer_block(sd, p, x, ctx, heads):
    """attention.py:421-541 (norm_type='layer_norm'); attn2/norm2 absent when cross_attention_dim is None."""
    C = x.shape[-1]
    n = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    x = attention(sd, p + "attn1.", n, None, heads) + x
    if (p + "attn2.to_q.weight") in sd:
        n = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        x = attention(sd, p + "attn2.", n, ctx, heads) + x
    n = F.layer_norm(x, (C,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    hg = F.linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])   # activations.py:117-123
    val, gate = hg.chunk(2, dim=-1)
    ff = F.linear(val * F.gelu(gate), sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    return ff + x


def transformer_2d(sd, p, x, ctx, heads, groups):
    """transformer_2d.py:479-527: GN(eps 1e-6) -> 1x1 proj_in -> tokens -> block -> 1x1 proj_out -> + residual."""
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    h = F.conv2d(h, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = basic_transformer_block(sd, p + "transformer_blocks.0.", h, ctx, heads)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    h = F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return h + res


def downsample(sd, p, x):
    """downsampling.py:132-149: conv3x3 stride 2 pad 1."""
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def upsample(sd, p, x, size=None):
    """upsampling.py:141-184: nearest x2 (or to explicit size) then conv3x3."""
    if size is None:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    else:
        x = F.interpolate(x, size=size, mode="nearest")
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)


def add_right(h, r):
    """The patched residual add (unet_2d_blocks.py:1303-1307 etc.): whole tensor on a square canvas,
    right-hand square `h[..., -H:]` on the width-concatenated canvas.  Returns a NEW tensor either way;
    aliasing effects of the reference's in-place form are handled explicitly by the callers."""
    if r is None:
        return h
    if h.shape[-1] == h.shape[-2]:
        return h + r
    h = h.clone()
    h[..., -h.shape[-2]:] = h[..., -h.shape[-2]:] + r
    return h


# ----------------------------------------------------------------------------------------------- trunks

def _down_blocks(sd, cfg, h, emb, ctx, adds, skips):
    """Down path shared by UNet (adds = residual list or None) and BlobNet (adds None).
    unet_2d_blocks.py:1241-1323 (CrossAttnDownBlock2D), :1378-1433 (DownBlock2D)."""
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        has_attn = i < nb - 1
        for j in range(cfg.layers_per_block):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}.", h, emb, cfg.norm_num_groups)
            if has_attn:
                h = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}.", h, ctx, cfg.num_heads, cfg.norm_num_groups)
            if adds is not None:
                h = add_right(h, adds.pop(0))
            skips.append(h)
        if i < nb - 1:
            h = downsample(sd, f"down_blocks.{i}.downsamplers.0.", h)
            if adds is not None:
                h = add_right(h, adds.pop(0))
            skips.append(h)
    return h


def _mid_block(sd, cfg, h, emb, ctx):
    """unet_2d_blocks.py:860-899."""
    h = resnet_block(sd, "mid_block.resnets.0.", h, emb, cfg.norm_num_groups)
    h = transformer_2d(sd, "mid_block.attentions.0.", h, ctx, cfg.num_heads, cfg.norm_num_groups)
    return resnet_block(sd, "mid_block.resnets.1.", h, emb, cfg.norm_num_groups)


def _up_blocks(sd, cfg, h, emb, ctx, adds, skips, collect, always_size):
    """unet_2d_blocks.py:2514-2624 (CrossAttnUpBlock2D), :2677-2765 (UpBlock2D).
    `always_size`: BlobNet forwards upsample_size for every non-final block (bn:894-895); the UNet only
    when the canvas is not a multiple of 2**num_upsamplers (unet_2d_condition.py:1116-1127) - for such
    sizes both give the same nearest-neighbour map, so the explicit size is always passed here."""
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        has_attn = i > 0
        n_res = cfg.layers_per_block + 1
        res = skips[-n_res:]
        del skips[-n_res:]
        for j in range(n_res):
            h = torch.cat([h, res.pop()], dim=1)
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}.", h, emb, cfg.norm_num_groups)
            if has_attn:
                h = transformer_2d(sd, f"up_blocks.{i}.attentions.{j}.", h, ctx, cfg.num_heads, cfg.norm_num_groups)
            if adds is not None:
                h = add_right(h, adds.pop(0))
            if collect is not None:
                collect.append(h)
        if i < nb - 1:
            size = skips[-1].shape[2:]
            h = upsample(sd, f"up_blocks.{i}.upsamplers.0.", h, size)
            if adds is not None:
                h = add_right(h, adds.pop(0))
            if collect is not None:
                collect.append(h)
    return h


def unet_forward(sd, cfg: NetConfig, sample, timestep, encoder_hidden_states,
                 down_adds: Optional[List[torch.Tensor]] = None, mid_add=None,
                 up_adds: Optional[List[torch.Tensor]] = None):
    """Patched UNet2DConditionModel.forward (unet_2d_condition.py:1039-1353).

    Aliasing quirk (unet_2d_condition.py:1213-1219): `down_block_res_samples = (sample,)` is captured BEFORE
    the first residual add.  On a non-square canvas the add is in place, so skip #0 carries the residual;
    on a square canvas `sample = sample + r` rebinds and skip #0 does NOT carry it."""
    down_adds = list(down_adds) if down_adds is not None else None
    up_adds = list(up_adds) if up_adds is not None else None
    emb = time_embed(sd, timestep, sample.shape[0], cfg)
    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    if down_adds is not None:
        square = h.shape[-1] == h.shape[-2]
        h_added = add_right(h, down_adds.pop(0))
        skips = [h if square else h_added]
        h = h_added
    else:
        skips = [h]
    h = _down_blocks(sd, cfg, h, emb, encoder_hidden_states, down_adds, skips)
    h = _mid_block(sd, cfg, h, emb, encoder_hidden_states)
    if mid_add is not None:
        h = add_right(h, mid_add)
    h = _up_blocks(sd, cfg, h, emb, encoder_hidden_states, up_adds, skips, None, False)
    h = F.group_norm(h, cfg.norm_num_groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], 1e-5)
    h = F.silu(h)
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def blobnet_forward(sd, cfg: NetConfig, sample, timestep, conditioning_scale: float):
    """BlobNetModel.forward (bn:720-945): returns (12 down residuals, mid residual, 15 up residuals),
    each = zero-conv 1x1 of the trunk feature, times conditioning_scale (bn:936-938)."""
    emb = time_embed(sd, timestep, sample.shape[0], cfg)
    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [h]
    h = _down_blocks(sd, cfg, h, emb, None, None, skips)
    down_feats = list(skips)
    h = _mid_block(sd, cfg, h, emb, None)
    mid_feat = h
    up_feats: List[torch.Tensor] = []
    _up_blocks(sd, cfg, h, emb, None, None, skips, up_feats, True)

    def zc(name, x):
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"]) * conditioning_scale

    down = [zc(f"blobnet_down_blocks.{i}", f) for i, f in enumerate(down_feats)]
    mid = zc("blobnet_mid_block", mid_feat)
    up = [zc(f"blobnet_up_blocks.{i}", f) for i, f in enumerate(up_feats)]
    return down, mid, up
