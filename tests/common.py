"""Shared helpers for the test-suite: tiny configs (must mirror tools/make_golden.py) and seeded inputs."""
import numpy as np
import torch

from blobctrl_amd import synth
from oracle.nets import NetConfig

TINY = dict(boc=(16, 32, 64, 64), groups=4, heads=2, ctx=16, feat=8, seed=7)


def g(seed, *shape):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32))


def tiny_cfgs():
    c = TINY
    ucfg = NetConfig(in_channels=5, out_channels=4, block_out_channels=c["boc"], num_heads=c["heads"],
                     norm_num_groups=c["groups"], cross_attention_dim=c["ctx"])
    bcfg = NetConfig(in_channels=4 + 1 + c["feat"], out_channels=0, block_out_channels=c["boc"], num_heads=c["heads"],
                     norm_num_groups=c["groups"], cross_attention_dim=None)
    return ucfg, bcfg


def tiny_weights():
    c = TINY
    us = synth.trunk_param_shapes(5, c["boc"], 2, c["ctx"], 4, blobnet=False)
    bs = synth.trunk_param_shapes(4 + 1 + c["feat"], c["boc"], 2, None, None, blobnet=True)
    return synth.synth_state_dict(us, c["seed"]), synth.synth_state_dict(bs, c["seed"] + 1)


def psnr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    peak = np.abs(b).max()
    return 10 * np.log10(peak * peak / max(mse, 1e-30))


# ---------------------------------------------------------------------------------------------------- tiny end-to-end pipeline
PIPE = dict(vae_boc=(32, 32, 64, 64), vae_groups=8, vae_seed=21, clip=dict(vocab=99, hidden=16, layers=2, inter=32, heads=2, seed=78),
            dino=dict(hidden=8, layers=2, heads=1, mlp_ratio=4, patch=14, grid=16, seed=98))


class FakeTokenizer:
    """Deterministic stand-in for CLIPTokenizer (vocabulary files are downloads): word -> 1 + (sum of code points mod 96), BOS 0,
    padding / EOS 98, `model_max_length` 77.  Same call contract the pipelines use (pipe:575-583, 651-657)."""
    model_max_length = 77
    added_tokens_encoder = {}

    def tokenize(self, text):
        return text.split()

    def batch_decode(self, *a, **k):
        return [""]

    def __call__(self, text, padding=None, max_length=None, truncation=None, return_tensors=None):
        if isinstance(text, str):
            text = [text]
        words = [[1 + (sum(map(ord, w)) % 96) for w in t.split()] for t in text]
        if max_length is None or padding == "longest":
            max_length = max(2, max(len(w) for w in words) + 2)
        ids = np.full((len(text), max_length), 98, np.int64)
        for i, w in enumerate(words):
            w = w[: max_length - 2]
            ids[i, 0] = 0
            ids[i, 1:1 + len(w)] = w
        out = type("Encoding", (), {})()
        out.input_ids = torch.from_numpy(ids)
        out.attention_mask = torch.ones_like(out.input_ids)
        return out


def tiny_pipeline_weights():
    """Seeded weights of the tiny VAE / CLIP text encoder / DINOv2 of the end-to-end pipeline fixture (both sides regenerate them)."""
    p = PIPE
    vae = synth.synth_state_dict(synth.vae_param_shapes(p["vae_boc"], 2, 4), p["vae_seed"])
    c = p["clip"]
    clip = synth.synth_state_dict(synth.clip_text_param_shapes(c["vocab"], c["hidden"], c["layers"], c["inter"], 77), c["seed"])
    d = p["dino"]
    dino = synth.synth_state_dict(synth.dinov2_param_shapes(d["hidden"], d["layers"], d["mlp_ratio"], d["patch"], d["grid"] ** 2), d["seed"])
    return vae, clip, dino


def pipeline_cases():
    """Keyword arguments of the end-to-end `__call__` cases in tests/golden/pipeline_call.npz (images are stored in the fixture)."""
    return {
        "unipc": dict(prompt=["a frog on a rock in a pond"], negative_prompt=None, num_inference_steps=4, guidance_scale=7.5,
                      blobnet_conditioning_scale=1.0, blobnet_control_guidance_start=0.0, blobnet_control_guidance_end=0.9,
                      num_images_per_prompt=1, seed=1248464818 % (2 ** 31), rng_seed=11, scheduler="unipc"),
        "ddim_neg2": dict(prompt=["a top hat beside a frog", "butterflies and flowers"], negative_prompt=["blurry", "low quality text"],
                          num_inference_steps=3, guidance_scale=5.0, blobnet_conditioning_scale=0.8,
                          blobnet_control_guidance_start=0.0, blobnet_control_guidance_end=1.0, num_images_per_prompt=1, seed=7,
                          rng_seed=12, scheduler="ddim"),
        "nocfg": dict(prompt="a rock", negative_prompt=None, num_inference_steps=2, guidance_scale=1.0, blobnet_conditioning_scale=1.0,
                      blobnet_control_guidance_start=0.0, blobnet_control_guidance_end=1.0, num_images_per_prompt=1, seed=3, rng_seed=13,
                      scheduler="unipc"),
    }


# ------------------------------------------------------------------------------------------------ block-level cases
# name -> (kind, parameters).  Weights come from blobctrl_amd.synth (PCG64 keyed by parameter name) on both sides; inputs from g().
BLOCK_SEED = 31
BLOCK_CASES = {
    "res_320_320": ("resnet", dict(cin=320, cout=320, B=2, H=8, W=16)),
    "res_320_640_shortcut": ("resnet", dict(cin=320, cout=640, B=2, H=8, W=16)),
    "res_2560_1280": ("resnet", dict(cin=2560, cout=1280, B=1, H=4, W=8)),
    "tfm_320_cross": ("transformer", dict(C=320, heads=8, ctx=768, B=2, H=8, W=16)),
    "tfm_320_self_only": ("transformer", dict(C=320, heads=8, ctx=None, B=2, H=8, W=16)),
    "up_scale2": ("upsample", dict(C=320, B=1, H=4, W=8, size=None)),
    "up_explicit_size": ("upsample", dict(C=320, B=1, H=3, W=6, size=(5, 12))),
    "down": ("downsample", dict(C=320, B=1, H=8, W=16)),
}
BLOCK_TIMESTEPS = (999, 981, 500, 1)


def block_param_shapes(kind, p):
    """Parameter names / shapes of ONE diffusers block, as its state_dict() spells them (pinned by tools/make_golden.py: strict load)."""
    from collections import OrderedDict
    sh = OrderedDict()
    if kind == "resnet":
        ci, co = p["cin"], p["cout"]
        sh["norm1.weight"], sh["norm1.bias"] = (ci,), (ci,)
        sh["conv1.weight"], sh["conv1.bias"] = (co, ci, 3, 3), (co,)
        sh["time_emb_proj.weight"], sh["time_emb_proj.bias"] = (co, 1280), (co,)
        sh["norm2.weight"], sh["norm2.bias"] = (co,), (co,)
        sh["conv2.weight"], sh["conv2.bias"] = (co, co, 3, 3), (co,)
        if ci != co:
            sh["conv_shortcut.weight"], sh["conv_shortcut.bias"] = (co, ci, 1, 1), (co,)
    elif kind == "transformer":
        C, ctx = p["C"], p["ctx"]
        sh["norm.weight"], sh["norm.bias"] = (C,), (C,)
        sh["proj_in.weight"], sh["proj_in.bias"] = (C, C, 1, 1), (C,)
        b = "transformer_blocks.0."
        sh[b + "norm1.weight"], sh[b + "norm1.bias"] = (C,), (C,)
        for n in ("to_q", "to_k", "to_v"):
            sh[b + f"attn1.{n}.weight"] = (C, C)
        sh[b + "attn1.to_out.0.weight"], sh[b + "attn1.to_out.0.bias"] = (C, C), (C,)
        if ctx is not None:
            sh[b + "norm2.weight"], sh[b + "norm2.bias"] = (C,), (C,)
            sh[b + "attn2.to_q.weight"] = (C, C)
            sh[b + "attn2.to_k.weight"], sh[b + "attn2.to_v.weight"] = (C, ctx), (C, ctx)
            sh[b + "attn2.to_out.0.weight"], sh[b + "attn2.to_out.0.bias"] = (C, C), (C,)
        sh[b + "norm3.weight"], sh[b + "norm3.bias"] = (C,), (C,)
        sh[b + "ff.net.0.proj.weight"], sh[b + "ff.net.0.proj.bias"] = (8 * C, C), (8 * C,)
        sh[b + "ff.net.2.weight"], sh[b + "ff.net.2.bias"] = (C, 4 * C), (C,)
        sh["proj_out.weight"], sh["proj_out.bias"] = (C, C, 1, 1), (C,)
    elif kind in ("upsample", "downsample"):
        sh["conv.weight"], sh["conv.bias"] = (p["C"], p["C"], 3, 3), (p["C"],)
    elif kind == "time":
        sh["linear_1.weight"], sh["linear_1.bias"] = (1280, 320), (1280,)
        sh["linear_2.weight"], sh["linear_2.bias"] = (1280, 1280), (1280,)
    return sh


def block_weights(name):
    from blobctrl_amd import synth
    kind, p = ("time", {}) if name == "time" else BLOCK_CASES[name]
    # (seed differs per case so that equally named parameters of two cases are not the same numbers)
    seed = BLOCK_SEED + (sorted(BLOCK_CASES).index(name) if name != "time" else 99)
    return synth.synth_state_dict(block_param_shapes(kind, p), seed)


def block_inputs(name):
    """(x NCHW, temb [B,1280] or None, context [B,7,ctx] or None) of a block case."""
    kind, p = BLOCK_CASES[name]
    i = sorted(BLOCK_CASES).index(name)
    C = p.get("cin", p.get("C"))
    x = g(500 + i, p["B"], C, p["H"], p["W"])
    temb = g(600 + i, p["B"], 1280) if kind == "resnet" else None
    ctx = g(700 + i, p["B"], 7, p["ctx"]) if kind == "transformer" and p["ctx"] is not None else None
    return x, temb, ctx
