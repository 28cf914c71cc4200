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
    # round 4: the block sizes rounds 3 / 4 built kernels for.  640 channels on a 32 x 64 map at B = 2 = 64 row blocks (the row-chain's
    # default threshold, split block end); 1280 channels on the 16 x 32 map (gemm_wreg.hip path).  The fixture keeps every `sub`-th
    # pixel of the reference output in both directions (the full maps would be 10 MB each).
    "tfm_640_cross": ("transformer", dict(C=640, heads=8, ctx=768, B=2, H=32, W=64, sub=8)),
    "tfm_640_self_only": ("transformer", dict(C=640, heads=8, ctx=None, B=2, H=32, W=64, sub=8)),
    "tfm_1280_cross": ("transformer", dict(C=1280, heads=8, ctx=768, B=2, H=16, W=32, sub=4)),
    "tfm_1280_self_only": ("transformer", dict(C=1280, heads=8, ctx=None, B=1, H=16, W=32, sub=4)),
}
# seed index of a case: the round-2 cases keep the index they had (their position in the sorted list of the original eight names)
_BLOCK_ORDER = ["down", "res_2560_1280", "res_320_320", "res_320_640_shortcut", "tfm_320_cross", "tfm_320_self_only", "up_explicit_size",
                "up_scale2", "tfm_640_cross", "tfm_640_self_only", "tfm_1280_cross", "tfm_1280_self_only"]
assert sorted(_BLOCK_ORDER) == sorted(BLOCK_CASES)
BLOCK_TIMESTEPS = (999, 981, 500, 1)


def block_index(name):
    return _BLOCK_ORDER.index(name)


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
    seed = BLOCK_SEED + (block_index(name) if name != "time" else 99)
    return synth.synth_state_dict(block_param_shapes(kind, p), seed)


def block_inputs(name):
    """(x NCHW, temb [B,1280] or None, context [B,7,ctx] or None) of a block case."""
    kind, p = BLOCK_CASES[name]
    i = block_index(name)
    C = p.get("cin", p.get("C"))
    x = g(500 + i, p["B"], C, p["H"], p["W"])
    temb = g(600 + i, p["B"], 1280) if kind == "resnet" else None
    ctx = g(700 + i, p["B"], 7, p["ctx"]) if kind == "transformer" and p["ctx"] is not None else None
    return x, temb, ctx


# ------------------------------------------------------------------------------------------------ on-disk model tree (construction path)
CLIP_TEST_WORDS = ["a", "frog", "sits", "on", "rock", "in", "pond", "with", "top", "hat", "beside", "it", "the", "butterflies", "and",
                   "flowers", "blurry", "low", "quality"]


def tiny_clip_vocab():
    """A synthetic CLIP BPE vocabulary: single characters (bare and word-final), a few merges, the two special tokens.  Small enough that
    every id stays below the tiny text encoder's 99-entry embedding table."""
    chars = sorted(set("".join(CLIP_TEST_WORDS)) | set(",.'"))
    merges = [("t", "h"), ("th", "e</w>"), ("o", "n</w>"), ("i", "n</w>"), ("a", "n"), ("an", "d</w>"), ("f", "r"), ("fr", "o"),
              ("fro", "g</w>"), ("r", "o"), ("ro", "c"), ("roc", "k</w>"), ("h", "a"), ("ha", "t</w>"), ("p", "o"), ("po", "n"),
              ("pon", "d</w>"), ("l", "o"), ("lo", "w</w>"), ("i", "t</w>"), ("e", "r")]
    toks = chars + [c + "</w>" for c in chars]
    for a, b in merges:
        toks.append(a + b)
    toks += ["<|startoftext|>", "<|endoftext|>"]
    return {t: i for i, t in enumerate(toks)}, merges


def write_model_tree(root, lora=True):
    """<root>/sd15/{unet (4-channel conv_in), vae, text_encoder, tokenizer, scheduler}, <root>/blobnet, <root>/dinov2,
    <root>/unet_lora/pytorch_lora_weights.safetensors: the tiny nets in the layout `scripts/blobctrl_inference.py:220-279` loads.
    Returns (paths, pieces) where pieces holds the in-memory state dicts / LoRA tensors for cross-checks."""
    import json
    import os
    from blobctrl_amd import checkpoint as ck
    c, p = TINY, PIPE
    root = str(root)
    mk = lambda *a: (os.makedirs(os.path.join(root, *a), exist_ok=True), os.path.join(root, *a))[1]
    base_unet = synth.synth_state_dict(synth.trunk_param_shapes(4, c["boc"], 2, c["ctx"], 4, blobnet=False), 5)
    _, bsd = tiny_weights()
    vsd, csd, dsd = tiny_pipeline_weights()
    d = mk("sd15", "unet")
    ck.write_safetensors(os.path.join(d, "diffusion_pytorch_model.safetensors"), base_unet)
    json.dump({"_class_name": "UNet2DConditionModel", "block_out_channels": list(c["boc"]), "attention_head_dim": c["heads"],
               "norm_num_groups": c["groups"], "cross_attention_dim": c["ctx"], "in_channels": 4, "out_channels": 4},
              open(os.path.join(d, "config.json"), "w"))
    d = mk("sd15", "vae")
    ck.write_safetensors(os.path.join(d, "diffusion_pytorch_model.safetensors"), vsd)
    json.dump({"_class_name": "AutoencoderKL", "block_out_channels": list(p["vae_boc"]), "norm_num_groups": p["vae_groups"],
               "layers_per_block": 2, "scaling_factor": 0.18215, "latent_channels": 4}, open(os.path.join(d, "config.json"), "w"))
    d = mk("sd15", "text_encoder")
    ck.write_safetensors(os.path.join(d, "model.safetensors"), csd)
    json.dump({"num_attention_heads": p["clip"]["heads"], "hidden_size": p["clip"]["hidden"], "layer_norm_eps": 1e-5,
               "num_hidden_layers": p["clip"]["layers"], "intermediate_size": p["clip"]["inter"], "vocab_size": p["clip"]["vocab"]},
              open(os.path.join(d, "config.json"), "w"))
    d = mk("sd15", "tokenizer")
    vocab, merges = tiny_clip_vocab()
    assert len(vocab) <= p["clip"]["vocab"], len(vocab)
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    with open(os.path.join(d, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    json.dump({"bos_token": "<|startoftext|>", "eos_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "unk_token": "<|endoftext|>"},
              open(os.path.join(d, "special_tokens_map.json"), "w"))
    json.dump({"model_max_length": 77, "tokenizer_class": "CLIPTokenizer", "pad_token": "<|endoftext|>", "bos_token": "<|startoftext|>",
               "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>"}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    d = mk("sd15", "scheduler")
    json.dump({"_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "beta_start": 0.00085, "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1,
               "trained_betas": None, "clip_sample": False}, open(os.path.join(d, "scheduler_config.json"), "w"))
    d = mk("blobnet")
    ck.write_safetensors(os.path.join(d, "diffusion_pytorch_model.safetensors"), bsd)
    json.dump({"block_out_channels": list(c["boc"]), "attention_head_dim": c["heads"], "norm_num_groups": c["groups"],
               "in_channels": 4, "conditioning_channels": 1 + c["feat"]}, open(os.path.join(d, "config.json"), "w"))
    d = mk("dinov2")
    ck.write_safetensors(os.path.join(d, "model.safetensors"), dsd)
    json.dump({"num_attention_heads": p["dino"]["heads"], "patch_size": p["dino"]["patch"], "layer_norm_eps": 1e-6,
               "hidden_size": p["dino"]["hidden"]}, open(os.path.join(d, "config.json"), "w"))
    json.dump({"crop_size": {"height": 224, "width": 224}, "size": {"shortest_edge": 256}, "image_mean": [0.485, 0.456, 0.406],
               "image_std": [0.229, 0.224, 0.225], "rescale_factor": 1 / 255, "image_processor_type": "BitImageProcessor"},
              open(os.path.join(d, "preprocessor_config.json"), "w"))
    wq = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"
    ff = "mid_block.attentions.0.transformer_blocks.0.ff.net.2"
    lora_t = {f"unet.{wq}.lora_A.weight": g(1, 4, c["boc"][0]) * 0.2, f"unet.{wq}.lora_B.weight": g(2, c["boc"][0], 4) * 0.2,
              "unet.conv_in.lora_A.weight": g(3, 4, 5, 3, 3) * 0.2, "unet.conv_in.lora_B.weight": g(4, c["boc"][0], 4, 1, 1) * 0.2,
              f"unet.{ff}.lora_A.weight": g(5, 4, 4 * c["boc"][-1]) * 0.1, f"unet.{ff}.lora_B.weight": g(6, c["boc"][-1], 4) * 0.1}
    if lora:
        d = mk("unet_lora")
        ck.write_safetensors(os.path.join(d, "pytorch_lora_weights.safetensors"), lora_t)
    paths = dict(sd15=os.path.join(root, "sd15"), blobnet=os.path.join(root, "blobnet"), dinov2=os.path.join(root, "dinov2"),
                 unet_lora=os.path.join(root, "unet_lora"))
    return paths, dict(unet4=base_unet, blobnet=bsd, vae=vsd, clip=csd, dino=dsd, lora=lora_t)


def set_plan(monkeypatch, **kv):
    """BC_PLAN for the plans recorded from here on (blobctrl_amd/options.py): planner options as key=value pairs, bools as 0 / 1."""
    monkeypatch.setenv("BC_PLAN", ",".join(f"{k}={int(v)}" for k, v in kv.items()))
