"""Parameter schema + deterministic synthetic weights for the SD-1.5 UNet / BlobNet / DINOv2 trunks.

There are no checkpoints offline, so benchmarks and parity tests run on seeded synthetic weights.  The
schema (names, shapes, order) is the reference's own `state_dict()` layout (SURVEY Appendix B;
/root/reference/blobctrl/models/blobnet.py:152-491, D/models/unets/unet_2d_condition.py:171-485) so that
a real `diffusion_pytorch_model.safetensors` drops in unchanged.  Values come from numpy PCG64 streams
keyed by the parameter NAME (stable across torch versions and across the build container / GPU box).
"""
import zlib
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch


def _resnet(shapes, p, cin, cout, temb):
    shapes[p + "norm1.weight"] = (cin,)
    shapes[p + "norm1.bias"] = (cin,)
    shapes[p + "conv1.weight"] = (cout, cin, 3, 3)
    shapes[p + "conv1.bias"] = (cout,)
    shapes[p + "time_emb_proj.weight"] = (cout, temb)
    shapes[p + "time_emb_proj.bias"] = (cout,)
    shapes[p + "norm2.weight"] = (cout,)
    shapes[p + "norm2.bias"] = (cout,)
    shapes[p + "conv2.weight"] = (cout, cout, 3, 3)
    shapes[p + "conv2.bias"] = (cout,)
    if cin != cout:
        shapes[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        shapes[p + "conv_shortcut.bias"] = (cout,)


def _transformer(shapes, p, c, ctx_dim):
    shapes[p + "norm.weight"] = (c,)
    shapes[p + "norm.bias"] = (c,)
    shapes[p + "proj_in.weight"] = (c, c, 1, 1)
    shapes[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    shapes[b + "norm1.weight"] = (c,)
    shapes[b + "norm1.bias"] = (c,)
    shapes[b + "attn1.to_q.weight"] = (c, c)
    shapes[b + "attn1.to_k.weight"] = (c, c)
    shapes[b + "attn1.to_v.weight"] = (c, c)
    shapes[b + "attn1.to_out.0.weight"] = (c, c)
    shapes[b + "attn1.to_out.0.bias"] = (c,)
    if ctx_dim is not None:
        shapes[b + "norm2.weight"] = (c,)
        shapes[b + "norm2.bias"] = (c,)
        shapes[b + "attn2.to_q.weight"] = (c, c)
        shapes[b + "attn2.to_k.weight"] = (c, ctx_dim)
        shapes[b + "attn2.to_v.weight"] = (c, ctx_dim)
        shapes[b + "attn2.to_out.0.weight"] = (c, c)
        shapes[b + "attn2.to_out.0.bias"] = (c,)
    shapes[b + "norm3.weight"] = (c,)
    shapes[b + "norm3.bias"] = (c,)
    shapes[b + "ff.net.0.proj.weight"] = (8 * c, c)
    shapes[b + "ff.net.0.proj.bias"] = (8 * c,)
    shapes[b + "ff.net.2.weight"] = (c, 4 * c)
    shapes[b + "ff.net.2.bias"] = (c,)
    shapes[p + "proj_out.weight"] = (c, c, 1, 1)
    shapes[p + "proj_out.bias"] = (c,)


def trunk_param_shapes(in_channels: int, block_out_channels: Tuple[int, ...], layers_per_block: int,
                       cross_attention_dim: Optional[int], out_channels: Optional[int],
                       blobnet: bool) -> "OrderedDict[str, tuple]":
    """Parameter names/shapes of the SD-1.5-topology trunk.  `blobnet=True` drops conv_norm_out/conv_out and
    attn2/norm2 (cross_attention_dim=None) and appends the 12 + 1 + 15 zero-convs (bn:336-349, 383-399, 480-491)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    boc = tuple(block_out_channels)
    nb = len(boc)
    temb = boc[0] * 4
    s["conv_in.weight"] = (boc[0], in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    s["time_embedding.linear_1.weight"] = (temb, boc[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    ch = boc[0]
    for i in range(nb):
        cin, cout = ch, boc[i]
        for j in range(layers_per_block):
            _resnet(s, f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout, temb)
            if i < nb - 1:
                _transformer(s, f"down_blocks.{i}.attentions.{j}.", cout, cross_attention_dim)
        if i < nb - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        ch = cout
    _resnet(s, "mid_block.resnets.0.", boc[-1], boc[-1], temb)
    _transformer(s, "mid_block.attentions.0.", boc[-1], cross_attention_dim)
    _resnet(s, "mid_block.resnets.1.", boc[-1], boc[-1], temb)
    rev = list(reversed(boc))
    prev = rev[0]
    for i in range(nb):
        cout = rev[i]
        cin_skip = rev[min(i + 1, nb - 1)]
        for j in range(layers_per_block + 1):
            skip_c = cin_skip if j == layers_per_block else cout
            res_in = prev if j == 0 else cout
            _resnet(s, f"up_blocks.{i}.resnets.{j}.", res_in + skip_c, cout, temb)
            if i > 0:
                _transformer(s, f"up_blocks.{i}.attentions.{j}.", cout, cross_attention_dim)
        if i < nb - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
        prev = cout
    if not blobnet:
        s["conv_norm_out.weight"] = (boc[0],)
        s["conv_norm_out.bias"] = (boc[0],)
        s["conv_out.weight"] = (out_channels, boc[0], 3, 3)
        s["conv_out.bias"] = (out_channels,)
    else:
        k = 0
        dch = [boc[0]]
        for i in range(nb):
            dch += [boc[i]] * layers_per_block
            if i < nb - 1:
                dch.append(boc[i])
        for k, c in enumerate(dch):
            s[f"blobnet_down_blocks.{k}.weight"] = (c, c, 1, 1)
            s[f"blobnet_down_blocks.{k}.bias"] = (c,)
        s["blobnet_mid_block.weight"] = (boc[-1], boc[-1], 1, 1)
        s["blobnet_mid_block.bias"] = (boc[-1],)
        uch = []
        for i in range(nb):
            uch += [rev[i]] * (layers_per_block + 1)
            if i < nb - 1:
                uch.append(rev[i])
        for k, c in enumerate(uch):
            s[f"blobnet_up_blocks.{k}.weight"] = (c, c, 1, 1)
            s[f"blobnet_up_blocks.{k}.bias"] = (c,)
    return s


def dinov2_param_shapes(hidden: int, layers: int, mlp_ratio: int, patch: int, num_pos: int) -> "OrderedDict[str, tuple]":
    """transformers `Dinov2Model.state_dict()` layout (call site pipe:690-703; ViT, pre-LN, LayerScale)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embeddings.cls_token"] = (1, 1, hidden)
    s["embeddings.mask_token"] = (1, hidden)
    s["embeddings.position_embeddings"] = (1, num_pos + 1, hidden)
    s["embeddings.patch_embeddings.projection.weight"] = (hidden, 3, patch, patch)
    s["embeddings.patch_embeddings.projection.bias"] = (hidden,)
    for i in range(layers):
        p = f"encoder.layer.{i}."
        s[p + "norm1.weight"] = (hidden,)
        s[p + "norm1.bias"] = (hidden,)
        for n in ("query", "key", "value"):
            s[p + f"attention.attention.{n}.weight"] = (hidden, hidden)
            s[p + f"attention.attention.{n}.bias"] = (hidden,)
        s[p + "attention.output.dense.weight"] = (hidden, hidden)
        s[p + "attention.output.dense.bias"] = (hidden,)
        s[p + "layer_scale1.lambda1"] = (hidden,)
        s[p + "norm2.weight"] = (hidden,)
        s[p + "norm2.bias"] = (hidden,)
        s[p + "mlp.fc1.weight"] = (hidden * mlp_ratio, hidden)
        s[p + "mlp.fc1.bias"] = (hidden * mlp_ratio,)
        s[p + "mlp.fc2.weight"] = (hidden, hidden * mlp_ratio)
        s[p + "mlp.fc2.bias"] = (hidden,)
        s[p + "layer_scale2.lambda1"] = (hidden,)
    s["layernorm.weight"] = (hidden,)
    s["layernorm.bias"] = (hidden,)
    return s


def clip_text_param_shapes(vocab: int = 49408, hidden: int = 768, layers: int = 12, intermediate: int = 3072,
                           max_pos: int = 77) -> "OrderedDict[str, tuple]":
    """transformers `CLIPTextModel.state_dict()` layout (call site pipe:599,668; defaults = SD-1.5's CLIP ViT-L/14 text tower)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["text_model.embeddings.token_embedding.weight"] = (vocab, hidden)
    s["text_model.embeddings.position_embedding.weight"] = (max_pos, hidden)
    for i in range(layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (hidden, hidden)
            s[p + f"self_attn.{n}.bias"] = (hidden,)
        s[p + "layer_norm1.weight"] = (hidden,)
        s[p + "layer_norm1.bias"] = (hidden,)
        s[p + "mlp.fc1.weight"] = (intermediate, hidden)
        s[p + "mlp.fc1.bias"] = (intermediate,)
        s[p + "mlp.fc2.weight"] = (hidden, intermediate)
        s[p + "mlp.fc2.bias"] = (hidden,)
        s[p + "layer_norm2.weight"] = (hidden,)
        s[p + "layer_norm2.bias"] = (hidden,)
    s["text_model.final_layer_norm.weight"] = (hidden,)
    s["text_model.final_layer_norm.bias"] = (hidden,)
    return s


def vae_param_shapes(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, in_channels=3,
                     out_channels=3) -> "OrderedDict[str, tuple]":
    """`AutoencoderKL.state_dict()` layout of the SD-1.5 VAE (D/models/autoencoders/autoencoder_kl.py:72-137, vae.py:47-348)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    boc = tuple(block_out_channels)
    nb = len(boc)

    def res(p, cin, cout):
        s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
        s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
        s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
        s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
        if cin != cout:
            s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + "conv_shortcut.bias"] = (cout,)

    def mid(p, c):
        s[p + "attentions.0.group_norm.weight"] = (c,); s[p + "attentions.0.group_norm.bias"] = (c,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[p + f"attentions.0.{n}.weight"] = (c, c); s[p + f"attentions.0.{n}.bias"] = (c,)
        res(p + "resnets.0.", c, c)
        res(p + "resnets.1.", c, c)

    s["encoder.conv_in.weight"] = (boc[0], in_channels, 3, 3); s["encoder.conv_in.bias"] = (boc[0],)
    ch = boc[0]
    for i in range(nb):
        for j in range(layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}.", ch if j == 0 else boc[i], boc[i])
        if i < nb - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (boc[i], boc[i], 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (boc[i],)
        ch = boc[i]
    mid("encoder.mid_block.", boc[-1])
    s["encoder.conv_norm_out.weight"] = (boc[-1],); s["encoder.conv_norm_out.bias"] = (boc[-1],)
    s["encoder.conv_out.weight"] = (2 * latent_channels, boc[-1], 3, 3); s["encoder.conv_out.bias"] = (2 * latent_channels,)
    rev = list(reversed(boc))
    s["decoder.conv_in.weight"] = (rev[0], latent_channels, 3, 3); s["decoder.conv_in.bias"] = (rev[0],)
    ch = rev[0]
    for i in range(nb):
        for j in range(layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}.", ch if j == 0 else rev[i], rev[i])
        if i < nb - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (rev[i], rev[i], 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (rev[i],)
        ch = rev[i]
    mid("decoder.mid_block.", rev[0])
    s["decoder.conv_norm_out.weight"] = (boc[0],); s["decoder.conv_norm_out.bias"] = (boc[0],)
    s["decoder.conv_out.weight"] = (out_channels, boc[0], 3, 3); s["decoder.conv_out.bias"] = (out_channels,)
    s["quant_conv.weight"] = (2 * latent_channels, 2 * latent_channels, 1, 1); s["quant_conv.bias"] = (2 * latent_channels,)
    s["post_quant_conv.weight"] = (latent_channels, latent_channels, 1, 1); s["post_quant_conv.bias"] = (latent_channels,)
    return s


def synth_tensor(name: str, shape: tuple, seed: int) -> np.ndarray:
    """One parameter, float32, from a PCG64 stream keyed by (seed, crc32(name)).

    norm scales ~ 1 + 0.1 N(0,1); biases / tokens / position embeddings ~ 0.02..0.05 N(0,1); LayerScale ~ 0.1..0.3;
    matrices / conv kernels ~ N(0, 1/fan_in) (variance preserving so activations stay O(1));
    BlobNet zero-convs get small NON-zero values (0.3/sqrt(fan_in)) so the BlobNet->UNet coupling is exercised
    (they are zero-initialised in the reference, bn:348-349 / bn:959-962)."""
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    n = int(np.prod(shape))
    x = rng.standard_normal(n, dtype=np.float32).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    if "lambda1" in name:
        return (0.2 + 0.05 * x).astype(np.float32)
    if "cls_token" in name or "mask_token" in name or "position_embeddings" in name:
        return (0.05 * x).astype(np.float32)
    if leaf == "bias":
        return (0.02 * x).astype(np.float32)
    if len(shape) == 1:                       # norm weight
        return (1.0 + 0.1 * x).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    gain = 0.3 if name.startswith("blobnet_") else 1.0
    return (x * (gain / np.sqrt(fan_in))).astype(np.float32)


def contractive_variant(usd: Dict[str, torch.Tensor], bsd: Dict[str, torch.Tensor], conv_out_scale: float = 0.25,
                        zero_conv_scale: float = 1.0):
    """Synthetic weights whose 50-step edit does not blow small differences up (SURVEY 7 hard part (ii)): the UNet's last convolution
    (and optionally BlobNet's zero-convs) scaled down, everything else unchanged.  With variance-preserving random weights the guided
    noise prediction has a Jacobian gain of 1.8 w.r.t. the latents at the first step (CFG 7.5 amplifies the branch difference), which
    compounds over 50 steps; scaled, eps stays O(1) and a perturbation grows only with the scheduler's own sqrt(abar_prev / abar_t)
    factors (tools/amplification_probe.py measures it).  Returns NEW dicts (the inputs are not modified)."""
    u = OrderedDict(usd)
    u["conv_out.weight"] = usd["conv_out.weight"] * conv_out_scale
    u["conv_out.bias"] = usd["conv_out.bias"] * conv_out_scale
    b = OrderedDict(bsd)
    if zero_conv_scale != 1.0:
        for k in bsd:
            if k.startswith("blobnet_"):
                b[k] = bsd[k] * zero_conv_scale
    return u, b


def synth_state_dict(shapes: "OrderedDict[str, tuple]", seed: int) -> Dict[str, torch.Tensor]:
    """Every parameter has its own stream (keyed by its name), so the tensors are generated on a thread pool (numpy's Generator
    releases the GIL) - same bits as a serial loop, a fraction of the 20 s that 860 M parameters take on one core."""
    items = list(shapes.items())
    if sum(int(np.prod(v)) for _, v in items) < (1 << 22):
        return OrderedDict((k, torch.from_numpy(synth_tensor(k, v, seed))) for k, v in items)
    import os
    from concurrent.futures import ThreadPoolExecutor
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=max(1, min(16, ncpu))) as ex:
        arrs = list(ex.map(lambda kv: synth_tensor(kv[0], kv[1], seed), items))
    return OrderedDict((k, torch.from_numpy(a)) for (k, _), a in zip(items, arrs))
