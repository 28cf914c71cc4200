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
