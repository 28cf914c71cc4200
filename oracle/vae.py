"""Oracle (TEST INFRASTRUCTURE): functional fp32 restatement of the SD-1.5 AutoencoderKL encode / decode.

Restates D/models/autoencoders/autoencoder_kl.py:249-326 (encode: encoder -> quant_conv -> DiagonalGaussianDistribution;
decode: post_quant_conv -> decoder), D/models/autoencoders/vae.py:47-183 (Encoder), :185-348 (Decoder), :767-789 (posterior
sample), D/models/unets/unet_2d_blocks.py:589-742 (UNetMidBlock2D: resnet, single-head attention with group_norm + residual,
resnet), D/models/resnet.py:320-373 with temb = None, D/models/downsampling.py:132-149 with padding = 0 (F.pad (0,1,0,1)),
D/models/upsampling.py:141-184.  The call sites on the BlobCtrl path are pipeline_blobnet.py:300-309 (encode_latents) and :1133.
"""
import torch
import torch.nn.functional as F


def _res(sd, p, x, groups):
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def _mid(sd, p, x, groups):
    x = _res(sd, p + "resnets.0.", x, groups)
    B, C, H, W = x.shape
    a = p + "attentions.0."
    n = F.group_norm(x, groups, sd[a + "group_norm.weight"], sd[a + "group_norm.bias"], 1e-6)
    t = n.view(B, C, H * W).transpose(1, 2)
    q = F.linear(t, sd[a + "to_q.weight"], sd[a + "to_q.bias"])
    k = F.linear(t, sd[a + "to_k.weight"], sd[a + "to_k.bias"])
    v = F.linear(t, sd[a + "to_v.weight"], sd[a + "to_v.bias"])
    s = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1)          # heads = C / attention_head_dim = 1
    o = F.linear(s @ v, sd[a + "to_out.0.weight"], sd[a + "to_out.0.bias"])
    x = o.transpose(1, 2).reshape(B, C, H, W) + x
    return _res(sd, p + "resnets.1.", x, groups)


def encode_moments(sd, x, groups=32, layers_per_block=2):
    """x [B,3,H,W] in [-1,1] -> moments [B, 2*Cz, H/8, W/8] (mean | logvar)."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    i = 0
    while f"encoder.down_blocks.{i}.resnets.0.norm1.weight" in sd:
        for j in range(layers_per_block):
            h = _res(sd, f"encoder.down_blocks.{i}.resnets.{j}.", h, groups)
        d = f"encoder.down_blocks.{i}.downsamplers.0.conv."
        if (d + "weight") in sd:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[d + "weight"], sd[d + "bias"], stride=2)
        i += 1
    h = _mid(sd, "encoder.mid_block.", h, groups)
    h = F.silu(F.group_norm(h, groups, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def sample_posterior(moments, noise, scale=0.18215):
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return (mean + torch.exp(0.5 * logvar) * noise) * scale


def decode(sd, z, groups=32, layers_per_block=2):
    """z [B,Cz,h,w] (already divided by the scaling factor) -> image [B,3,8h,8w]."""
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _mid(sd, "decoder.mid_block.", h, groups)
    i = 0
    while f"decoder.up_blocks.{i}.resnets.0.norm1.weight" in sd:
        for j in range(layers_per_block + 1):
            h = _res(sd, f"decoder.up_blocks.{i}.resnets.{j}.", h, groups)
        u = f"decoder.up_blocks.{i}.upsamplers.0.conv."
        if (u + "weight") in sd:
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[u + "weight"], sd[u + "bias"], padding=1)
        i += 1
    h = F.silu(F.group_norm(h, groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
