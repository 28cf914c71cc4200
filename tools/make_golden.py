#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on CPU fp32.

Runs only in the build container (needs /root/reference).  Nothing from the reference is copied: the fixtures hold
inputs and expected outputs only.  Weights are NOT stored - both sides regenerate them from
`blobctrl_amd.synth` (numpy PCG64 keyed by parameter name), and this script checks `load_state_dict(strict=True)`
so the parameter schema in `blobctrl_amd/synth.py` is pinned against the reference classes too.

Import recipe: vendored diffusers first, FLAX_WEIGHTS_NAME shim, stub modules for cv2/torchvision.
"""
import json
import os
import sys
import types
import importlib.machinery

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path[:0] = ["/root/reference/diffusers/src", "/root/reference", REPO]

import numpy as np
import torch

import transformers.utils
transformers.utils.FLAX_WEIGHTS_NAME = "flax_model.msgpack"
import diffusers  # noqa: E402  (vendored 0.30.0)
for _name in ["cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
              "matplotlib", "matplotlib.cm"]:
    try:
        __import__(_name)
    except Exception:
        _m = types.ModuleType(_name)
        _m.__spec__ = importlib.machinery.ModuleSpec(_name, None)
        _m.__path__ = []
        sys.modules[_name] = _m

from diffusers import UNet2DConditionModel, DDIMScheduler, UniPCMultistepScheduler  # noqa: E402
from blobctrl.models.blobnet import BlobNetModel  # noqa: E402
from blobctrl.utils.utils import splat_features  # noqa: E402
from blobctrl.pipelines.pipeline_blobnet import StableDiffusionBlobNetPipeline  # noqa: E402
import scripts.blobctrl_inference as ref_inf  # noqa: E402

from blobctrl_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)
torch.set_num_threads(8)

# ------------------------------------------------------------------------------------------------ tiny configs
TINY = dict(boc=(16, 32, 64, 64), groups=4, heads=2, ctx=16, feat=8, seed=7)
SD_SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                steps_offset=1)


def build_tiny():
    c = TINY
    unet = UNet2DConditionModel(in_channels=5, out_channels=4, block_out_channels=c["boc"], norm_num_groups=c["groups"],
                                attention_head_dim=c["heads"], cross_attention_dim=c["ctx"], layers_per_block=2)
    # NB: the reference script leaves unet.config.in_channels = 4 (inf:233-249); only conv_in has 5 inputs.
    blob = BlobNetModel(in_channels=4, conditioning_channels=1 + c["feat"], block_out_channels=c["boc"],
                        norm_num_groups=c["groups"], attention_head_dim=c["heads"], cross_attention_dim=None,
                        layers_per_block=2)
    us = synth.trunk_param_shapes(5, c["boc"], 2, c["ctx"], 4, blobnet=False)
    bs = synth.trunk_param_shapes(4 + 1 + c["feat"], c["boc"], 2, None, None, blobnet=True)
    assert set(us.keys()) == set(unet.state_dict().keys()), "UNet schema mismatch"
    assert set(bs.keys()) == set(blob.state_dict().keys()), "BlobNet schema mismatch"
    unet.load_state_dict(synth.synth_state_dict(us, c["seed"]), strict=True)
    blob.load_state_dict(synth.synth_state_dict(bs, c["seed"] + 1), strict=True)
    return unet.eval(), blob.eval()


def check_full_schema():
    """Pin the full-size schema (names + shapes) on the meta device: no memory, no arithmetic."""
    with torch.device("meta"):
        unet = UNet2DConditionModel(in_channels=5, out_channels=4, cross_attention_dim=768, attention_head_dim=8)
        blob = BlobNetModel(in_channels=4, conditioning_channels=1025, cross_attention_dim=None, attention_head_dim=8)
    us = synth.trunk_param_shapes(5, (320, 640, 1280, 1280), 2, 768, 4, blobnet=False)
    bs = synth.trunk_param_shapes(1029, (320, 640, 1280, 1280), 2, None, None, blobnet=True)
    for sh, m, n in ((us, unet, "unet"), (bs, blob, "blobnet")):
        ref = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert set(ref.keys()) == set(sh.keys()), n
        assert all(ref[k] == tuple(sh[k]) for k in ref), n
        print(f"full-size {n}: {len(ref)} tensors, {sum(int(np.prod(v)) for v in ref.values())/1e6:.1f} M params - schema OK")


def g(seed, *shape):
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32))


# ------------------------------------------------------------------------------------------------ 1. splat
def golden_splat():
    cases = []
    demo = "/root/reference/assets/results/demo"
    for name in sorted(os.listdir(demo)):
        st = json.load(open(os.path.join(demo, name, "state", "state.json")))
        for idx in (0, -1):
            cases.append((name, st["ellipse_lists"][idx][0], 512, 512, 64, 64))
    cases.append(("move_hat_cli", [[361.1067, 367.8526], [85.4812, 103.6543], 87.3739], 512, 512, 64, 64))
    cases.append(("hires_768", [[541.66, 551.78], [128.22, 155.48], 87.3739], 768, 768, 96, 96))
    cases.append(("nonsquare", [[300.0, 120.0], [60.0, 150.0], 33.0], 640, 384, 48, 80))
    cases.append(("degenerate", [[256.0, 256.0], [1e-5, 1e-5], 0.0], 512, 512, 64, 64))     # app:1384 'add' start ellipse
    out = {}
    meta = []
    for i, (name, ell, W, H, h, w) in enumerate(cases):
        mean, cov = ref_inf.get_gs_from_ellipse(ell)
        nm, nc = ref_inf.normalize_gs(mean, cov, W, H)
        blob = ref_inf.get_blob_dict_from_norm_gs(nm, nc)
        score = splat_features(**blob, score_size=(h, w), return_d_score=True)          # [1,2,h,w] float64
        out[f"score_{i}"] = score.numpy()
        out[f"mean_{i}"] = nm
        out[f"cov_{i}"] = nc
        meta.append(dict(name=name, ellipse=ell, W=W, H=H, h=h, w=w))
    # size < 0.5 branch (ut:165-172)
    blob = ref_inf.get_blob_dict_from_norm_gs(out["mean_0"], out["cov_0"])
    blob["sizes"] = torch.tensor([[0.2]])
    out["score_absent"] = splat_features(**blob, score_size=(64, 64), return_d_score=True).numpy()
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "splat.npz"), **out)
    s = out["score_%d" % [m["name"] for m in meta].index("move_hat_cli")]
    print("splat: fg.sum=%.6f fg.max=%.6f" % (s[0, 1].sum(), s[0, 1].max()))


# ------------------------------------------------------------------------------------------------ 2. nets
def golden_nets():
    unet, blob = build_tiny()
    c = TINY
    out = {}
    # "odd": canvas 10 x 24 (latent 10 x 12) - feature maps 10x24 -> 5x12 -> 3x6 -> 2x3: the explicit-size upsampling branch
    for tag, (h, w) in (("wide", (8, 16)), ("square", (8, 8)), ("odd", (10, 24))):
        B = 2
        x_b = g(11, B, 4 + 1 + c["feat"], h, w)
        t = torch.tensor(981)
        down, mid, up = blob(x_b, t, conditioning_scale=0.8, return_dict=False)
        out[f"{tag}_blob_in"] = x_b.numpy()
        for i, r in enumerate(down):
            out[f"{tag}_down_{i}"] = r.numpy()
        out[f"{tag}_mid"] = mid.numpy()
        for i, r in enumerate(up):
            out[f"{tag}_up_{i}"] = r.numpy()
        x_u = g(12, B, 5, h, w)
        ehs = g(13, B, 7, c["ctx"])
        sq = (lambda r: r[..., -r.shape[-2]:])
        eps = unet(x_u, t, encoder_hidden_states=ehs,
                   down_block_add_samples=[sq(r).clone() for r in down], mid_block_add_sample=sq(mid).clone(),
                   up_block_add_samples=[sq(r).clone() for r in up], return_dict=False)[0]
        eps_plain = unet(x_u, t, encoder_hidden_states=ehs, return_dict=False)[0]
        out[f"{tag}_unet_in"] = x_u.numpy()
        out[f"{tag}_ehs"] = ehs.numpy()
        out[f"{tag}_eps"] = eps.numpy()
        out[f"{tag}_eps_plain"] = eps_plain.numpy()
    out["timestep"] = np.array(981)
    np.savez_compressed(os.path.join(OUT, "nets_tiny.npz"), **out)
    print("nets: eps std %.4f, residual-coupled delta %.4f" % (out["wide_eps"].std(),
                                                               np.abs(out["wide_eps"] - out["wide_eps_plain"]).max()))


# ------------------------------------------------------------------------------------------------ 3. schedulers
def golden_schedulers():
    out = {}
    for n in (5, 20, 50):
        uni = UniPCMultistepScheduler(**SD_SCHED)
        ddim = DDIMScheduler(**SD_SCHED, clip_sample=False, set_alpha_to_one=False)
        for name, sch in (("unipc", uni), ("ddim", ddim)):
            sch.set_timesteps(n)
            out[f"{name}_{n}_timesteps"] = sch.timesteps.numpy()
            x = g(21, 1, 4, 8, 8)
            xs = [x.numpy()]
            for i, t in enumerate(sch.timesteps):
                eps = g(100 + i, 1, 4, 8, 8)
                x = sch.step(eps, t, x, return_dict=False)[0]
                xs.append(x.numpy())
            out[f"{name}_{n}_traj"] = np.stack(xs)
        out[f"unipc_{n}_sigmas"] = uni.sigmas.numpy()
    np.savez_compressed(os.path.join(OUT, "schedulers.npz"), **out)
    print("schedulers: unipc50 t0..2", out["unipc_50_timesteps"][:3], "ddim50", out["ddim_50_timesteps"][:3])


# ------------------------------------------------------------------------------------------------ 4. pipeline loop
class _FakeVaeCfg:
    scaling_factor = 0.18215
    block_out_channels = (1, 1, 1, 1)


def golden_loop():
    """Drive the reference pipeline's own __call__ loop code on the tiny nets by calling its methods directly:
    construct_blobnet_input + blobnet + unet + CFG + scheduler.step exactly as pipe:1025-1102 does."""
    unet, blob = build_tiny()
    c = TINY
    h = w = 8
    B = 1
    out = {}
    pipe = StableDiffusionBlobNetPipeline.__new__(StableDiffusionBlobNetPipeline)   # only stateless helpers are used
    for sname in ("unipc", "ddim"):
        for steps, (gs, ge) in ((5, (0.0, 1.0)), (6, (0.0, 0.67))):
            sch = (UniPCMultistepScheduler(**SD_SCHED) if sname == "unipc"
                   else DDIMScheduler(**SD_SCHED, clip_sample=False, set_alpha_to_one=False))
            sch.set_timesteps(steps)
            latents = g(31, B, 4, h, w) * sch.init_noise_sigma
            prompt = g(32, 2 * B, 7, c["ctx"])
            fg_lat = (g(33, 1, 4, h, w) * 0.18215 * 5).repeat(2 * B, 1, 1, 1)
            bg_lat = (g(34, 1, 4, h, w) * 0.18215 * 5).repeat(2 * B, 1, 1, 1)
            ell = [[40.0, 42.0], [20.0, 30.0], 25.0]
            mean, cov = ref_inf.get_gs_from_ellipse(ell)
            nm, nc = ref_inf.normalize_gs(mean, cov, 64, 64)
            gs_score = splat_features(**ref_inf.get_blob_dict_from_norm_gs(nm, nc), score_size=(h, w),
                                      return_d_score=True)
            bg_s, fg_s = gs_score.unbind(dim=1)
            bg_s = bg_s.unsqueeze(1).repeat(2 * B, 1, 1, 1).float()
            fg_s = fg_s.unsqueeze(1).repeat(2 * B, 1, 1, 1).float()
            dino = g(35, 1, 1, c["feat"])
            feats = pipe.splat_features_from_scores(fg_s, dino.repeat(2 * B, 1, 1), size=h, channels_last=False)
            keep = [1.0 - float(i / steps < gs or (i + 1) / steps > ge) for i in range(steps)]
            eps_trace = []
            for i, t in enumerate(sch.timesteps):
                lmi = sch.scale_model_input(torch.cat([latents] * 2), t)
                cond = 1.0 * keep[i]
                bi = pipe.construct_blobnet_input(lmi, fg_s, fg_lat, feats, background=False)
                d, m, u = blob(bi, t, conditioning_scale=cond, return_dict=False)
                ui = pipe.construct_blobnet_input(lmi, bg_s, bg_lat, background=True)
                npred = unet(ui, t, encoder_hidden_states=prompt,
                             down_block_add_samples=[x[..., -x.shape[-2]:] for x in d],
                             mid_block_add_sample=m[..., -m.shape[-2]:],
                             up_block_add_samples=[x[..., -x.shape[-2]:] for x in u], return_dict=False)[0]
                b_, c_, h_, w_ = npred.shape
                npred = npred[..., :h_, w_ // 2:]
                nu, nt = npred.chunk(2)
                npred = nu + 7.5 * (nt - nu)
                eps_trace.append(npred.numpy())
                latents = sch.step(npred, t, latents, return_dict=False)[0]
            tag = f"{sname}_{steps}"
            out[f"{tag}_final"] = latents.numpy()
            out[f"{tag}_eps"] = np.stack(eps_trace)
            out[f"{tag}_window"] = np.array([gs, ge])
            if sname == "unipc" and steps == 5:
                out["gs_score"] = gs_score.numpy()
                out["ellipse"] = np.array([40.0, 42.0, 20.0, 30.0, 25.0])
            print(f"loop {tag}: final std {latents.std():.4f}")
    np.savez_compressed(os.path.join(OUT, "loop_tiny.npz"), **out)


def golden_fullsize_loop(variants=None):
    """The north star's own parity statement: a FREE-RUNNING 50-step 512x512 edit on the full-size SD-1.5 UNet +
    BlobNet (reference classes, CPU fp32, the loop body of pipe:1025-1102 as golden_loop drives it), bench.py's synthetic weights and
    inputs, CFG 7.5, guidance window [0, 0.9] (45 BlobNet-active steps + 5 UNet-only).  Only the final latents and three intermediate
    checkpoints per run are stored (4 x 64 KB).  ~25-40 min per run on 8 cores.  Variants: "<scheduler>:<conv_out_scale>", default
    all four of {unipc, ddim} x {1.0 (the benchmark's weights), 0.3 (synth.contractive_variant)}; runs already in the file are kept.
    tools/amplification_probe.py measured how much the edit amplifies a perturbation of its start latents: 2.6 (DDIM) / 3.2 (UniPC)
    on the benchmark's weights, 1.3 / 1.5 at conv_out_scale 0.3."""
    import time
    import bench
    variants = variants or ["unipc:1.0", "ddim:1.0", "unipc:0.3", "ddim:0.3"]
    path = os.path.join(OUT, "loop_fullsize.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    h = w = 64
    steps, ge = 50, 0.9
    inp = bench.synth_inputs(h, w)
    usd0, bsd0 = bench.synth_weights()
    with torch.device("meta"):
        unet = UNet2DConditionModel(in_channels=5, out_channels=4, cross_attention_dim=768, attention_head_dim=8)
        blob = BlobNetModel(in_channels=4, conditioning_channels=1025, cross_attention_dim=None, attention_head_dim=8)
    pipe = StableDiffusionBlobNetPipeline.__new__(StableDiffusionBlobNetPipeline)
    ell = [[361.1067, 367.8526], [85.4812, 103.6543], 87.3739]
    mean, cov = ref_inf.get_gs_from_ellipse(ell)
    nm, nc = ref_inf.normalize_gs(mean, cov, 512, 512)
    gs_score = splat_features(**ref_inf.get_blob_dict_from_norm_gs(nm, nc), score_size=(h, w), return_d_score=True)
    bg_s, fg_s = gs_score.unbind(dim=1)
    bg_s, fg_s = bg_s.unsqueeze(1).repeat(2, 1, 1, 1).float(), fg_s.unsqueeze(1).repeat(2, 1, 1, 1).float()
    fg_lat, bg_lat = inp["fg"].repeat(2, 1, 1, 1), inp["bg"].repeat(2, 1, 1, 1)
    feats = pipe.splat_features_from_scores(fg_s, inp["dino"].repeat(2, 1, 1), size=h, channels_last=False)
    loaded = None
    for var in variants:
        sname, scale = var.split(":")
        scale = float(scale)
        tag = f"{sname}_s{scale:g}"
        if f"{tag}_final" in out:
            print("have", tag)
            continue
        if loaded != scale:
            usd, bsd = synth.contractive_variant(usd0, bsd0, conv_out_scale=scale) if scale != 1.0 else (usd0, bsd0)
            unet.load_state_dict(usd, strict=True, assign=True)
            blob.load_state_dict(bsd, strict=True, assign=True)
            unet.eval(); blob.eval()
            loaded = scale
        sch = (UniPCMultistepScheduler(**SD_SCHED) if sname == "unipc" else DDIMScheduler(**SD_SCHED, clip_sample=False, set_alpha_to_one=False))
        sch.set_timesteps(steps)
        latents = inp["latents"] * sch.init_noise_sigma
        keep = [1.0 - float(i / steps < 0.0 or (i + 1) / steps > ge) for i in range(steps)]
        t0 = time.time()
        for i, t in enumerate(sch.timesteps):
            lmi = sch.scale_model_input(torch.cat([latents] * 2), t)
            bi = pipe.construct_blobnet_input(lmi, fg_s, fg_lat, feats, background=False)
            d, m, u = blob(bi, t, conditioning_scale=1.0 * keep[i], return_dict=False)
            ui = pipe.construct_blobnet_input(lmi, bg_s, bg_lat, background=True)
            npred = unet(ui, t, encoder_hidden_states=inp["prompt"], down_block_add_samples=[x[..., -x.shape[-2]:] for x in d],
                         mid_block_add_sample=m[..., -m.shape[-2]:], up_block_add_samples=[x[..., -x.shape[-2]:] for x in u],
                         return_dict=False)[0]
            npred = npred[..., :, npred.shape[-1] // 2:]
            nu, nt = npred.chunk(2)
            npred = nu + 7.5 * (nt - nu)
            latents = sch.step(npred, t, latents, return_dict=False)[0]
            if i + 1 in (1, 10, 25, 40) and scale == 1.0:          # (checkpoints only for the benchmark's own weights: 64 KB each)
                out[f"{tag}_x{i + 1}"] = latents.numpy().copy()
            print(f"{tag} step {i + 1}/{steps} |x| max {latents.abs().max():.2f} ({time.time() - t0:.0f} s)", flush=True)
        out[f"{tag}_final"] = latents.numpy().copy()
        out["window_end"] = np.float32(ge)
        np.savez_compressed(path, **out)                      # after every run: a later interruption keeps the finished ones
        print("wrote", tag, "->", path)


def _ref_loop(unet, blob, pipe, sch, steps, ge, latents, prompt, fg, bg, gs_score, dino, strength, h, keep_at=()):
    """The loop body of pipe:1025-1102 on the reference classes (one request, CFG 7.5); returns the final latents and checkpoints."""
    import time
    bg_s, fg_s = gs_score.unbind(dim=1)
    bg_s, fg_s = bg_s.unsqueeze(1).repeat(2, 1, 1, 1).float(), fg_s.unsqueeze(1).repeat(2, 1, 1, 1).float()
    fg_lat, bg_lat = fg.repeat(2, 1, 1, 1), bg.repeat(2, 1, 1, 1)
    feats = pipe.splat_features_from_scores(fg_s, dino.repeat(2, 1, 1), size=h, channels_last=False)
    sch.set_timesteps(steps)
    latents = latents * sch.init_noise_sigma
    keep = [1.0 - float(i / steps < 0.0 or (i + 1) / steps > ge) for i in range(steps)]
    cps, t0 = {}, time.time()
    for i, t in enumerate(sch.timesteps):
        lmi = sch.scale_model_input(torch.cat([latents] * 2), t)
        bi = pipe.construct_blobnet_input(lmi, fg_s, fg_lat, feats, background=False)
        d, m, u = blob(bi, t, conditioning_scale=float(strength) * keep[i], return_dict=False)
        ui = pipe.construct_blobnet_input(lmi, bg_s, bg_lat, background=True)
        npred = unet(ui, t, encoder_hidden_states=prompt, down_block_add_samples=[x[..., -x.shape[-2]:] for x in d],
                     mid_block_add_sample=m[..., -m.shape[-2]:], up_block_add_samples=[x[..., -x.shape[-2]:] for x in u],
                     return_dict=False)[0]
        npred = npred[..., :, npred.shape[-1] // 2:]
        nu, nt = npred.chunk(2)
        latents = sch.step(nu + 7.5 * (nt - nu), t, latents, return_dict=False)[0]
        if i + 1 in keep_at:
            cps[i + 1] = latents.numpy().copy()
        print(f"   step {i + 1}/{steps} |x| max {latents.abs().max():.2f} ({time.time() - t0:.0f} s)", flush=True)
    return latents.numpy().copy(), cps


def golden_config_loops(which=None):
    """multi-step FREE-RUNNING reference runs at the other BASELINE configurations (only C2 had one).
    (i) `c5`: 768 x 768 (canvas 96 x 192), batch 1, 10 DDIM steps, window [0, 0.9] on bench.py's synthetic weights / inputs;
    (ii) `c3_r0` / `c3_r1`: requests 0 (move, strength 1.0) and 1 (`remove`: strength 0.0, gs_score = (1, 0), inf:175-188) of the
    mixed-operation batch of tests/test_fullsize_loop_gpu.py::test_c3_*, each alone (the reference runs one edit per call), 10 UniPC
    steps.  Stored: final latents + checkpoints after steps 1 and 5.  ~45 s per 768^2 step and ~15 s per 512^2 step on 8 cores."""
    import bench
    which = which or ["c5", "c3_r0", "c3_r1"]
    path = os.path.join(OUT, "loop_configs.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    with torch.device("meta"):
        unet = UNet2DConditionModel(in_channels=5, out_channels=4, cross_attention_dim=768, attention_head_dim=8)
        blob = BlobNetModel(in_channels=4, conditioning_channels=1025, cross_attention_dim=None, attention_head_dim=8)
    usd, bsd = bench.synth_weights()
    unet.load_state_dict(usd, strict=True, assign=True)
    blob.load_state_dict(bsd, strict=True, assign=True)
    unet.eval(); blob.eval()
    pipe = StableDiffusionBlobNetPipeline.__new__(StableDiffusionBlobNetPipeline)
    steps, ge = 10, 0.9

    def ref_score(ell, W, H, h, w):
        mean, cov = ref_inf.get_gs_from_ellipse(ell)
        nm, nc = ref_inf.normalize_gs(mean, cov, W, H)
        return splat_features(**ref_inf.get_blob_dict_from_norm_gs(nm, nc), score_size=(h, w), return_d_score=True)

    for tag in which:
        if f"{tag}_final" in out:
            print("have", tag)
            continue
        print("running", tag, flush=True)
        if tag == "c5":
            h = w = 96
            inp = bench.synth_inputs(h, w)
            s = (8 * w) / 512.0
            score = ref_score([[361.1067 * s, 367.8526 * s], [85.4812 * s, 103.6543 * s], 87.3739], 8 * w, 8 * h, h, w)
            sch = DDIMScheduler(**SD_SCHED, clip_sample=False, set_alpha_to_one=False)
            final, cps = _ref_loop(unet, blob, pipe, sch, steps, ge, inp["latents"], inp["prompt"], inp["fg"], inp["bg"], score,
                                   inp["dino"], 1.0, h, keep_at=(1, 5))
        else:
            b = int(tag[-1])
            B, h, w = 8, 64, 64
            ells = [[[200.0 + 30 * i, 180.0 + 25 * i], [60.0 + 6 * i, 90.0 - 4 * i], 20.0 * i] for i in range(B)]
            strengths = [1.0, 0.0, 1.0, 1.2, 0.0, 1.0, 0.6, 1.0]
            score = ref_score(ells[b], 512, 512, h, w)
            if strengths[b] == 0.0:
                score[:, 0], score[:, 1] = 1.0, 0.0
            fg, bg = g(201, B, 4, h, w) * 0.18215 * 5, g(202, B, 4, h, w) * 0.18215 * 5
            dino, lat = g(203, B, 1, 1024), g(204, B, 4, h, w)
            neg, pos = g(205, 1, 77, 768).repeat(B, 1, 1), g(206, B, 77, 768)
            sch = UniPCMultistepScheduler(**SD_SCHED)
            final, cps = _ref_loop(unet, blob, pipe, sch, steps, ge, lat[b:b + 1], torch.cat([neg[b:b + 1], pos[b:b + 1]]),
                                   fg[b:b + 1], bg[b:b + 1], score, dino[b:b + 1], strengths[b], h, keep_at=(1, 5))
        out[f"{tag}_final"] = final
        for k, v in cps.items():
            out[f"{tag}_x{k}"] = v
        out["steps"], out["window_end"] = np.int32(steps), np.float32(ge)
        np.savez_compressed(path, **out)
        print("wrote", tag, "->", path, flush=True)


def golden_cpu_step_seconds():
    """ONE denoise step of the REAL reference (vendored diffusers UNet + BlobNet classes at CFG batch 2, fp32,
    pipe:1043-1090) timed beside the oracle's `noise_pred_step` on the same cores, same weights and inputs, interleaved (one warm-up
    each, then alternating, best of 2): tests/golden/cpu_step_seconds.json.  bench.py's cpu_baseline (kind "port") quotes the ratio
    so that the port is shown to cost what the reference's own path costs (it was 3x slower while the oracle materialised the
    attention scores instead of calling F.scaled_dot_product_attention)."""
    import time
    import bench
    from oracle.nets import NetConfig
    from oracle.pipeline import noise_pred_step
    h = w = 64
    inp = bench.synth_inputs(h, w)
    usd, bsd = bench.synth_weights()
    with torch.device("meta"):
        unet = UNet2DConditionModel(in_channels=5, out_channels=4, cross_attention_dim=768, attention_head_dim=8)
        blob = BlobNetModel(in_channels=4, conditioning_channels=1025, cross_attention_dim=None, attention_head_dim=8)
    unet.load_state_dict(usd, strict=True, assign=True)
    blob.load_state_dict(bsd, strict=True, assign=True)
    unet.eval(); blob.eval()
    pipe = StableDiffusionBlobNetPipeline.__new__(StableDiffusionBlobNetPipeline)
    ell = [[361.1067, 367.8526], [85.4812, 103.6543], 87.3739]
    mean, cov = ref_inf.get_gs_from_ellipse(ell)
    nm, nc = ref_inf.normalize_gs(mean, cov, 512, 512)
    gs_score = splat_features(**ref_inf.get_blob_dict_from_norm_gs(nm, nc), score_size=(h, w), return_d_score=True)
    bg_s, fg_s = gs_score.unbind(dim=1)
    bg_s, fg_s = bg_s.unsqueeze(1).repeat(2, 1, 1, 1).float(), fg_s.unsqueeze(1).repeat(2, 1, 1, 1).float()
    fg_lat, bg_lat = inp["fg"].repeat(2, 1, 1, 1), inp["bg"].repeat(2, 1, 1, 1)
    feats = pipe.splat_features_from_scores(fg_s, inp["dino"].repeat(2, 1, 1), size=h, channels_last=False)
    ucfg, bcfg = NetConfig(in_channels=5, cross_attention_dim=768), NetConfig(in_channels=1029, cross_attention_dim=None)
    x = inp["latents"]

    def ref_step(t):
        lmi = torch.cat([x] * 2)
        bi = pipe.construct_blobnet_input(lmi, fg_s, fg_lat, feats, background=False)
        d, m, u = blob(bi, t, conditioning_scale=1.0, return_dict=False)
        ui = pipe.construct_blobnet_input(lmi, bg_s, bg_lat, background=True)
        npred = unet(ui, t, encoder_hidden_states=inp["prompt"], down_block_add_samples=[r[..., -r.shape[-2]:] for r in d],
                     mid_block_add_sample=m[..., -m.shape[-2]:], up_block_add_samples=[r[..., -r.shape[-2]:] for r in u],
                     return_dict=False)[0]
        npred = npred[..., :, npred.shape[-1] // 2:]
        nu, nt = npred.chunk(2)
        return nu + 7.5 * (nt - nu)

    def port_step(t):
        return noise_pred_step(usd, ucfg, bsd, bcfg, x, t, inp["prompt"], fg_lat, bg_lat, fg_s, bg_s, feats, 1.0, 7.5)

    times = {"reference": [], "port": []}
    outs = {}
    for rnd in range(3):                                   # round 0 = warm-up of both
        for name, fn in (("reference", ref_step), ("port", port_step)):
            t0 = time.perf_counter()
            outs[name] = fn(torch.tensor(981))
            dt = time.perf_counter() - t0
            if rnd:
                times[name].append(round(dt, 3))
            print(f"round {rnd} {name}: {dt:.2f} s", flush=True)
    rel = float((outs["port"] - outs["reference"]).abs().max() / outs["reference"].abs().max())
    doc = dict(threads=torch.get_num_threads(), cpus=len(os.sched_getaffinity(0)), resolution="512x512",
               what="one BlobNet-active denoise step, BlobNet AND UNet at CFG batch 2, fp32, after one warm-up step; best of 2",
               reference_s=min(times["reference"]), port_s=min(times["port"]), samples=times,
               port_vs_reference_ratio=round(min(times["port"]) / min(times["reference"]), 3),
               guided_eps_max_abs_rel_port_vs_reference=rel)
    with open(os.path.join(OUT, "cpu_step_seconds.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc))


# ------------------------------------------------------------------------------------------------ 5. dinov2
def golden_dinov2():
    from transformers import Dinov2Config, Dinov2Model
    cfg = Dinov2Config(hidden_size=64, num_hidden_layers=3, num_attention_heads=4, image_size=70, patch_size=14,
                       mlp_ratio=4)
    model = Dinov2Model(cfg).eval()
    shapes = synth.dinov2_param_shapes(64, 3, 4, 14, 25)
    assert set(shapes.keys()) == set(model.state_dict().keys()), "dinov2 schema mismatch"
    model.load_state_dict(synth.synth_state_dict(shapes, 99), strict=True)
    out = {}
    for tag, s in (("native", 70), ("interp", 56)):
        x = g(41, 2, 3, s, s)
        out[f"{tag}_in"] = x.numpy()
        out[f"{tag}_pooled"] = model(pixel_values=x).pooler_output.numpy()
    np.savez_compressed(os.path.join(OUT, "dinov2_tiny.npz"), **out)
    print("dinov2: pooled std %.4f" % out["native_pooled"].std())


def golden_blob_edit():
    """Pure-geometry functions of scripts/blobctrl_app.py (the app itself needs gradio / cv2 / SAM and loads models at import, so
    only the wanted function definitions are taken out of its source with `ast` and executed here; cv2.boundingRect is the one
    OpenCV call among them and is replaced by its definition - min / max of the non-zero pixels)."""
    import ast
    from PIL import Image
    src = open("/root/reference/scripts/blobctrl_app.py").read()
    want = ["normalize_ellipse", "composite_mask_and_image", "is_point_in_ellipse", "calculate_ellipse_vertices", "move_ellipse",
            "resize_blob_func", "rotate_blob_func", "get_object_region_from_mask"]
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(body) == len(want)

    class _Gr:
        warnings = []

        @staticmethod
        def Warning(msg):
            _Gr.warnings.append(msg)

    class _Cv2:
        @staticmethod
        def boundingRect(m):
            ys, xs = np.nonzero(m)
            return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)
    ns = {"np": np, "Image": Image, "gr": _Gr, "cv2": _Cv2}
    exec(compile(ast.Module(body=body, type_ignores=[]), "blobctrl_app_subset", "exec"), ns)
    rng = np.random.Generator(np.random.PCG64(17))
    out = {"ellipses": [], "normalize": [], "vertices": [], "inside": [], "move": [], "resize": [], "rotate": []}
    ells = [((256.0, 256.0), (120.0, 80.0), 30.0), ((361.1067, 367.8526), (85.4812, 103.6543), 87.3739),
            ((40.0, 470.0), (300.0, 40.0), 135.0), ((256.0, 256.0), (1e-5, 1e-5), 0.0), ((100.0, 90.0), (60.0, 60.0), 179.0)]
    for e in ells:
        out["ellipses"].append([list(e[0]), list(e[1]), e[2]])
        out["normalize"].append([float(v) for v in ns["normalize_ellipse"](e, 512, 384)])
        out["vertices"].append(ns["calculate_ellipse_vertices"](e).tolist())
        pts = rng.uniform(0, 512, size=(12, 2))
        out["inside"].append([[float(p[0]), float(p[1]), bool(ns["is_point_in_ellipse"](p, e))] for p in pts])
        tp = rng.uniform(0, 512, size=(3, 2)).tolist()
        m = ns["move_ellipse"](e, tp)
        out["move"].append({"points": tp, "out": [list(m[0]), list(m[1]), m[2]]})
        for rt in (0, 1, 2):
            for f in (1.0, 0.5, 1.7, 6.0, 0.05):
                _Gr.warnings = []
                r, rf = ns["resize_blob_func"](e, f, 512, 512, rt)
                out["resize"].append({"ellipse": len(out["ellipses"]) - 1, "factor": f, "type": rt,
                                      "out": [list(r[0]), list(r[1]), r[2]], "factor_out": float(rf),
                                      "too_big": any("too big" in w for w in _Gr.warnings),
                                      "too_small": any("too small" in w for w in _Gr.warnings)})
        for deg in (0.0, 45.0, 170.0, -30.0):
            r, _ = ns["rotate_blob_func"](e, deg)
            out["rotate"].append({"ellipse": len(out["ellipses"]) - 1, "deg": deg, "out": [list(r[0]), list(r[1]), r[2]]})
    img = rng.integers(0, 256, size=(24, 32, 3)).astype(np.uint8)
    mask = np.zeros((24, 32), np.uint8)
    mask[5:14, 9:21] = 255
    mask[10:18, 15:25] = 128
    mask3 = np.stack([mask, mask, np.zeros_like(mask)], -1)
    comp = np.array(ns["composite_mask_and_image"](mask, img, masked_color=[255, 255, 255]))
    comp3 = np.array(ns["composite_mask_and_image"](mask3, img, masked_color=[0, 0, 0]))
    obj = np.array(ns["get_object_region_from_mask"](mask, img))
    out["image"] = img.tolist()
    out["mask"] = mask.tolist()
    out["composite_white"] = comp.tolist()
    out["composite_rgbmask_black"] = comp3.tolist()
    out["object_region"] = obj.tolist()
    with open(os.path.join(OUT, "blob_edit.json"), "w") as f:
        json.dump(out, f)
    print("blob edit:", len(out["resize"]), "resize cases,", sum(r["too_big"] for r in out["resize"]), "too big,",
          sum(r["too_small"] for r in out["resize"]), "too small")


def golden_clip_text():
    """CLIP text tower (transformers, third-party; call site pipe:599-611 incl. the clip_skip branch)."""
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=99, hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                         max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    model = CLIPTextModel(cfg).eval()
    shapes = synth.clip_text_param_shapes(99, 64, 3, 128, 77)
    # the on-disk SD-1.5 checkpoint (transformers 4.49 layout) prefixes every key with "text_model."; the transformers installed
    # here (5.x) dropped the prefix - same tensors
    have = {k for k in model.state_dict().keys() if "position_ids" not in k}
    pre = "" if any(k.startswith("text_model.") for k in have) else "text_model."
    assert set(shapes.keys()) == {pre + k for k in have}, "clip schema mismatch"
    sd = synth.synth_state_dict(shapes, 77)
    model.load_state_dict({k[len(pre):]: v for k, v in sd.items()}, strict=False)
    rng = np.random.Generator(np.random.PCG64(5))
    ids = torch.from_numpy(rng.integers(1, 98, size=(2, 77)).astype(np.int64))
    ids[:, 0] = 0
    ids[0, 20:] = 98                                       # eos padding pattern
    out = {"ids": ids.numpy()}
    with torch.no_grad():
        r = model(ids, output_hidden_states=True)
    out["last_hidden_state"] = r[0].numpy()
    fln = model.text_model.final_layer_norm if hasattr(model, "text_model") else model.final_layer_norm
    out["clip_skip_1"] = fln(r.hidden_states[-2]).detach().numpy()                                    # pipe:604-611
    short = ids[:, :16].clone()
    with torch.no_grad():
        out["short_ids"] = short.numpy()
        out["short_last_hidden_state"] = model(short)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "clip_text_tiny.npz"), **out)
    print("clip text: last_hidden std %.4f" % out["last_hidden_state"].std())


# ------------------------------------------------------------------------------------------------ 6. VAE
def golden_vae():
    from diffusers import AutoencoderKL
    with torch.device("meta"):
        full = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                             up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512),
                             layers_per_block=2, latent_channels=4, norm_num_groups=32, sample_size=512)
    sh = synth.vae_param_shapes()
    ref = {k: tuple(v.shape) for k, v in full.state_dict().items()}
    assert set(ref) == set(sh) and all(ref[k] == tuple(sh[k]) for k in ref), "VAE schema mismatch"
    print(f"full-size VAE: {len(ref)} tensors, {sum(int(np.prod(v)) for v in ref.values())/1e6:.1f} M params - schema OK")
    boc = (32, 32, 64, 64)
    vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=boc, layers_per_block=2,
                        latent_channels=4, norm_num_groups=8, sample_size=32).eval()
    tsh = synth.vae_param_shapes(boc, 2, 4)
    assert set(tsh) == set(vae.state_dict().keys())
    vae.load_state_dict(synth.synth_state_dict(tsh, 21), strict=True)
    out = {}
    x = g(51, 2, 3, 32, 48).clamp(-1, 1)
    post = vae.encode(x).latent_dist
    out["img"] = x.numpy()
    out["moments"] = post.parameters.numpy()
    z = g(52, 2, 4, 4, 6)
    out["z"] = z.numpy()
    out["decoded"] = vae.decode(z, return_dict=False)[0].numpy()
    gen = torch.Generator().manual_seed(5)
    noise = torch.randn(post.mean.shape, generator=gen)
    gen = torch.Generator().manual_seed(5)
    out["noise"] = noise.numpy()
    out["sample"] = post.sample(generator=gen).numpy()
    np.savez_compressed(os.path.join(OUT, "vae_tiny.npz"), **out)
    print("vae: decoded std %.4f moments std %.4f" % (out["decoded"].std(), out["moments"].std()))


def golden_lora_keys():
    """Key renaming + alpha defaults of the reference LoRA loader (peft itself is absent here, so only the parts that live in the
    vendored diffusers tree are pinned: convert_unet_state_dict_to_peft and get_peft_kwargs)."""
    from diffusers.utils.state_dict_utils import convert_unet_state_dict_to_peft
    from diffusers.utils.peft_utils import get_peft_kwargs
    mods = ["down_blocks.0.attentions.0.transformer_blocks.0.attn1", "mid_block.attentions.0.transformer_blocks.0.attn2",
            "up_blocks.3.attentions.2.transformer_blocks.0.attn1.processor"]
    keys = []
    for m in mods:
        for proj in ("to_q", "to_k", "to_v", "to_out"):
            for d in ("down", "up"):
                keys.append(f"{m}.{proj}_lora.{d}.weight")
    keys += ["conv_in.lora.down.weight", "conv_in.lora.up.weight", "down_blocks.1.resnets.0.conv1.lora.down.weight",
             "down_blocks.1.resnets.0.conv1.lora.up.weight", "down_blocks.0.attentions.0.proj_in.lora_A.weight",
             "down_blocks.0.attentions.0.proj_in.lora_B.weight", "up_blocks.1.attentions.0.transformer_blocks.0.ff.net.2.lora.down.weight",
             "up_blocks.1.attentions.0.transformer_blocks.0.ff.net.2.lora.up.weight",
             "mid_block.attentions.0.transformer_blocks.0.attn1.to_out.0.lora_A.weight"]
    conv = convert_unet_state_dict_to_peft({k: i for i, k in enumerate(keys)})
    out = {"pairs": [[k, nk] for k, nk in zip(keys, conv.keys())]}
    # alpha / rank defaults: ranks in file order 4, 8, 4, 4 without alphas; then with alphas
    peft_sd = {"a.lora_A.weight": 0, "a.lora_B.weight": 0, "b.lora_A.weight": 0, "b.lora_B.weight": 0,
               "c.lora_A.weight": 0, "c.lora_B.weight": 0}
    ranks = {"a.lora_B.weight": 8, "b.lora_B.weight": 4, "c.lora_B.weight": 4}
    out["kwargs_no_alpha"] = {k: v for k, v in get_peft_kwargs(ranks, None, peft_sd).items() if k != "target_modules"}
    alphas = {"a.lora_A.weight.alpha": 16.0, "b.lora_A.weight.alpha": 4.0, "c.lora_A.weight.alpha": 4.0}
    out["kwargs_alpha"] = {k: v for k, v in get_peft_kwargs(ranks, alphas, peft_sd).items() if k != "target_modules"}
    with open(os.path.join(OUT, "lora_keys.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("lora keys:", len(out["pairs"]), "pairs;", out["kwargs_no_alpha"], out["kwargs_alpha"])


def golden_lora_merge():
    """The adapter arithmetic as the REFERENCE tree states it (peft is absent, but the vendored diffusers keeps its
    own LoRA layers): D/models/lora.py LoRALinearLayer / LoRAConv2dLayer forward (runtime branch: base(x) + lora_scale * up(down(x))
    [* network_alpha / rank]) and LoRACompatibleLinear / LoRACompatibleConv._fuse_lora (w + lora_scale * (alpha / r) * up . down).
    Linear and Conv2d 3x3, with and without network_alpha.  Stored: the factors, the input, the reference's runtime output, its fused
    weight and the fused layer's output."""
    import warnings
    from diffusers.models.lora import LoRACompatibleConv, LoRACompatibleLinear, LoRAConv2dLayer, LoRALinearLayer
    out = {}
    warnings.simplefilter("ignore")
    for tag, alpha in (("noalpha", None), ("alpha", 2.0)):
        # Linear 24 -> 20, rank 4
        lin = LoRACompatibleLinear(24, 20, bias=True)
        lin.weight.data = g(901, 20, 24) * 0.2
        lin.bias.data = g(902, 20) * 0.1
        ll = LoRALinearLayer(24, 20, rank=4, network_alpha=alpha)
        ll.down.weight.data = g(903, 4, 24) * 0.3
        ll.up.weight.data = g(904, 20, 4) * 0.3
        lin.set_lora_layer(ll)
        x = g(905, 3, 7, 24)
        y_rt = lin(x, 0.7)
        w0 = lin.weight.data.clone()
        lin._fuse_lora(lora_scale=0.7)
        y_f = lin(x)
        out.update({f"lin_{tag}_w": w0.numpy(), f"lin_{tag}_b": lin.bias.data.numpy(), f"lin_{tag}_down": ll.down.weight.data.numpy(),
                    f"lin_{tag}_up": ll.up.weight.data.numpy(), f"lin_{tag}_x": x.numpy(), f"lin_{tag}_y_runtime": y_rt.numpy(),
                    f"lin_{tag}_w_fused": lin.weight.data.numpy(), f"lin_{tag}_y_fused": y_f.numpy()})
        # Conv2d 3x3 6 -> 8, rank 4 (down is 3x3, up is 1x1: lora.py:275-278)
        cv = LoRACompatibleConv(6, 8, kernel_size=3, padding=1, bias=True)
        cv.weight.data = g(911, 8, 6, 3, 3) * 0.2
        cv.bias.data = g(912, 8) * 0.1
        cl = LoRAConv2dLayer(6, 8, rank=4, kernel_size=(3, 3), stride=(1, 1), padding=1, network_alpha=alpha)
        cl.down.weight.data = g(913, 4, 6, 3, 3) * 0.3
        cl.up.weight.data = g(914, 8, 4, 1, 1) * 0.3
        cv.set_lora_layer(cl)
        xc = g(915, 2, 6, 5, 9)
        y_rt = cv(xc, 0.7)
        w0 = cv.weight.data.clone()
        cv._fuse_lora(lora_scale=0.7)
        y_f = cv(xc)
        out.update({f"conv_{tag}_w": w0.numpy(), f"conv_{tag}_b": cv.bias.data.numpy(), f"conv_{tag}_down": cl.down.weight.data.numpy(),
                    f"conv_{tag}_up": cl.up.weight.data.numpy(), f"conv_{tag}_x": xc.numpy(), f"conv_{tag}_y_runtime": y_rt.numpy(),
                    f"conv_{tag}_w_fused": cv.weight.data.numpy(), f"conv_{tag}_y_fused": y_f.numpy()})
        for k in ("lin", "conv"):
            d = np.abs(out[f"{k}_{tag}_y_runtime"] - out[f"{k}_{tag}_y_fused"]).max()
            assert d < 1e-5, (k, tag, d)
    out["lora_scale"] = np.float32(0.7)
    out["network_alpha"] = np.float32(2.0)
    np.savez_compressed(os.path.join(OUT, "lora_merge.npz"), **out)
    print("lora merge:", sorted(out)[:4], "...")


def golden_blob_edit_cv():
    """OpenCV outputs the REFERENCE ITSELF holds (cv2 is absent here): for every demo under assets/results/demo the SAM mask
    (ori_result_gallery_2.png), the filled ellipse mask (ori_result_gallery_3.png = cv2.ellipse(.., 255, -1), app:717) and the state's
    first ellipse (= cv2.fitEllipse(convexHull(contours)) enlarged by the initial factor 1.05, app:382-389,902).  Stored as packed
    bit masks + the ellipse parameters; `fit_pinned` lists the demos whose first ellipse is the unedited fit x 1.05."""
    from PIL import Image
    root = "/root/reference/assets/results/demo"
    out = {}
    names = sorted(d for d in os.listdir(root) if os.path.exists(os.path.join(root, d, "state", "state.json")))
    for name in names:
        st = json.load(open(os.path.join(root, name, "state", "state.json")))
        e = st["ellipse_lists"][0][0]
        out[f"{name}_ellipse0"] = np.array([e[0][0], e[0][1], e[1][0], e[1][1], e[2]], np.float64)
        for idx, key in ((2, "mask"), (3, "filled")):
            im = np.array(Image.open(os.path.join(root, name, "ori_result_gallery", f"ori_result_gallery_{idx}.png")).convert("RGB")).astype(np.int32)
            out[f"{name}_{key}"] = np.packbits(im.sum(-1) > 3 * 127)               # (the gallery files went through a lossy codec)
            out[f"{name}_shape"] = np.array(im.shape[:2])
    out["names"] = np.array(names)
    out["enlarge_factor"] = np.float64(1.05)
    np.savez_compressed(os.path.join(OUT, "blob_edit_cv.npz"), **out)
    print("blob_edit_cv:", names, os.path.getsize(os.path.join(OUT, "blob_edit_cv.npz")) // 1024, "KB")


def golden_pipeline_call():
    """The REFERENCE pipeline's own `__call__` (pipe:743-1166) end to end on tiny components: PIL images + prompt strings in, latents
    (and one decoded image) out - tokenizer stand-in, CLIP, VAE encode (global-generator posterior samples), DINOv2 processor + model,
    rank-1 splat, BlobNet + patched UNet loop, scheduler, VAE decode, VaeImageProcessor.postprocess."""
    from PIL import Image
    from diffusers import AutoencoderKL
    from transformers import BitImageProcessor, CLIPTextConfig, CLIPTextModel, Dinov2Config, Dinov2Model
    from tests.common import PIPE, FakeTokenizer, pipeline_cases, tiny_pipeline_weights
    c = TINY
    unet5, blob = build_tiny()
    # the script's construction (inf:229-249): a 4-channel UNet whose conv_in is replaced by a 5-channel one; config.in_channels stays 4
    unet = UNet2DConditionModel(in_channels=4, out_channels=4, block_out_channels=c["boc"], norm_num_groups=c["groups"],
                                attention_head_dim=c["heads"], cross_attention_dim=c["ctx"], layers_per_block=2)
    unet.conv_in = torch.nn.Conv2d(5, unet.conv_in.out_channels, kernel_size=3, stride=1, padding=1)
    unet.load_state_dict(unet5.state_dict(), strict=True)
    unet.eval()
    assert unet.config.in_channels == 4
    vsd, csd, dsd = tiny_pipeline_weights()
    vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=PIPE["vae_boc"], layers_per_block=2,
                        latent_channels=4, norm_num_groups=PIPE["vae_groups"], sample_size=32).eval()
    vae.load_state_dict(vsd, strict=True)
    pc = PIPE["clip"]
    te = CLIPTextModel(CLIPTextConfig(vocab_size=pc["vocab"], hidden_size=pc["hidden"], intermediate_size=pc["inter"],
                                      num_hidden_layers=pc["layers"], num_attention_heads=pc["heads"], max_position_embeddings=77,
                                      hidden_act="quick_gelu", layer_norm_eps=1e-5)).eval()
    have = {k for k in te.state_dict().keys() if "position_ids" not in k}
    pre = "" if any(k.startswith("text_model.") for k in have) else "text_model."
    assert set(csd.keys()) == {pre + k for k in have}, "clip schema mismatch"
    te.load_state_dict({k[len(pre):]: v for k, v in csd.items()}, strict=False)
    pd = PIPE["dino"]
    dino = Dinov2Model(Dinov2Config(hidden_size=pd["hidden"], num_hidden_layers=pd["layers"], num_attention_heads=pd["heads"],
                                    image_size=14 * pd["grid"], patch_size=pd["patch"], mlp_ratio=pd["mlp_ratio"])).eval()
    assert set(dsd.keys()) == set(dino.state_dict().keys()), "dinov2 schema mismatch"
    dino.load_state_dict(dsd, strict=True)
    # facebook/dinov2-* preprocessor_config.json values (external knowledge, like the ViT-L sizes): what AutoImageProcessor resolves to
    proc = BitImageProcessor(do_resize=True, size={"shortest_edge": 256}, resample=3, do_center_crop=True,
                             crop_size={"height": 224, "width": 224}, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                             image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225], do_convert_rgb=True)
    rng = np.random.Generator(np.random.PCG64(3))
    # smooth-ish images (random low-res blocks upsampled) so that resize filters matter but nothing saturates
    def img(h, w):
        lo = rng.integers(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)
        return np.kron(lo, np.ones((8, 8, 1), np.uint8)) // 2 + rng.integers(0, 128, (h, w, 3)).astype(np.uint8)
    out = {"fg": img(64, 64), "bg": img(64, 64)}
    ell = [[40.0, 42.0], [20.0, 30.0], 25.0]
    mean, cov = ref_inf.get_gs_from_ellipse(ell)
    nm, nc = ref_inf.normalize_gs(mean, cov, 64, 64)
    gs = splat_features(**ref_inf.get_blob_dict_from_norm_gs(nm, nc), score_size=(8, 8), return_d_score=True)   # [1,2,8,8] f64
    out["gs_score"] = gs.numpy()
    # image-processor pins (host code on our side): DINOv2 preprocessing of non-square images, VAE preprocessing with a resize
    from diffusers.image_processor import VaeImageProcessor
    odd = Image.fromarray(img(72, 104))
    out["odd"] = np.array(odd)
    out["odd_dino_pixels"] = proc.preprocess(images=odd, do_resize=True, return_tensors="pt", do_convert_rgb=True).pixel_values.numpy()
    out["odd_vae_pixels"] = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True).preprocess(odd, height=64, width=96).numpy()
    for name, kw in pipeline_cases().items():
        kw = dict(kw)
        sch = (UniPCMultistepScheduler(**SD_SCHED) if kw.pop("scheduler") == "unipc"
               else DDIMScheduler(**SD_SCHED, clip_sample=False, set_alpha_to_one=False))
        pipe = StableDiffusionBlobNetPipeline(vae=vae, unet=unet, tokenizer=FakeTokenizer(), text_encoder=te, blobnet=blob,
                                              scheduler=sch, safety_checker=None, dinov2_processor=proc, dinov2=dino,
                                              requires_safety_checker=False)
        pipe.set_progress_bar_config(disable=True)
        # record what the reference's own front ends hand to its loop (the LOOP-ENTRY tensors): prompt embeddings, image latents in
        # call order (fg, then bg), DINOv2 feature - so that the GPU test can bound the front ends and the loop separately
        entry = {"lat": []}

        def tap(fn, key):
            def wrapped(*a, **k_):
                r_ = fn(*a, **k_)
                if key == "lat":
                    entry["lat"].append(r_.detach().clone())
                else:
                    entry[key] = r_
                return r_
            return wrapped
        pipe.encode_prompt = tap(pipe.encode_prompt, "prompt")
        pipe.encode_latents = tap(pipe.encode_latents, "lat")
        pipe.encode_image_dinov2 = tap(pipe.encode_image_dinov2, "dino")
        seed, rng_seed = kw.pop("seed"), kw.pop("rng_seed")
        common = dict(fg_image=Image.fromarray(out["fg"]), bg_image=Image.fromarray(out["bg"]), gs_score=gs, height=64, width=64, **kw)
        torch.manual_seed(rng_seed)           # the VAE posterior samples come from the GLOBAL generator (pipe:304)
        r = pipe(generator=torch.Generator().manual_seed(seed), output_type="latent", **common)
        out[f"{name}_latents"] = r.images.numpy()
        pe, ne = entry["prompt"]
        out[f"{name}_entry_prompt_embeds"] = pe.numpy()
        if ne is not None:
            out[f"{name}_entry_negative_prompt_embeds"] = ne.numpy()
        out[f"{name}_entry_fg_latents"] = entry["lat"][0][:1].numpy()
        out[f"{name}_entry_bg_latents"] = entry["lat"][1][:1].numpy()
        out[f"{name}_entry_dino"] = entry["dino"].numpy()
        if name == "unipc":
            torch.manual_seed(rng_seed)
            r = pipe(generator=torch.Generator().manual_seed(seed), output_type="np", **common)
            out[f"{name}_image_np"] = np.asarray(r.images)
        print(f"pipeline __call__ {name}: latents std {out[f'{name}_latents'].std():.4f}")
    np.savez_compressed(os.path.join(OUT, "pipeline_call.npz"), **out)


def golden_blocks():
    """single diffusers blocks at SD-1.5 widths on small feature maps (resnet.py:320-373, transformer_2d.py:479-527,
    upsampling.py:141-184, downsampling.py:132-149, embeddings.py:27-78 + 576-588).  Only the OUTPUTS are stored (fp32); weights and
    inputs are regenerated from seeds on both sides (tests/common.py block_weights / block_inputs)."""
    from diffusers.models.resnet import ResnetBlock2D
    from diffusers.models.transformers.transformer_2d import Transformer2DModel
    from diffusers.models.upsampling import Upsample2D
    from diffusers.models.downsampling import Downsample2D
    from diffusers.models.embeddings import Timesteps, TimestepEmbedding
    from tests.common import BLOCK_CASES, BLOCK_TIMESTEPS, block_inputs, block_weights
    out = {}
    for name, (kind, p) in BLOCK_CASES.items():
        x, temb, ctx = block_inputs(name)
        if kind == "resnet":       # as the UNet blocks build it (unet_2d_blocks.py: eps=resnet_eps=1e-5, groups=32, temb 1280)
            m = ResnetBlock2D(in_channels=p["cin"], out_channels=p["cout"], temb_channels=1280, eps=1e-5, groups=32)
        elif kind == "transformer":
            m = Transformer2DModel(num_attention_heads=p["heads"], attention_head_dim=p["C"] // p["heads"], in_channels=p["C"],
                                   num_layers=1, cross_attention_dim=p["ctx"], norm_num_groups=32)
        elif kind == "upsample":
            m = Upsample2D(p["C"], use_conv=True, out_channels=p["C"])
        else:
            m = Downsample2D(p["C"], use_conv=True, out_channels=p["C"], padding=1, name="op")
        m.eval()
        sd = block_weights(name)
        assert set(sd.keys()) == set(m.state_dict().keys()), (name, set(sd.keys()) ^ set(m.state_dict().keys()))
        m.load_state_dict(sd, strict=True)
        if kind == "resnet":
            y = m(x, temb)
        elif kind == "transformer":
            y = m(x, encoder_hidden_states=ctx, return_dict=False)[0]
        elif kind == "upsample":
            y = m(x, output_size=p["size"])
        else:
            y = m(x)
        sub = p.get("sub", 1)
        out[name] = y.numpy().astype(np.float32)[:, :, ::sub, ::sub]
        print(f"block {name}: out {tuple(y.shape)} std {float(y.std()):.4f} stored {out[name].shape}")
    proj, emb = Timesteps(320, True, 0), TimestepEmbedding(320, 1280).eval()
    emb.load_state_dict(block_weights("time"), strict=True)
    t = torch.tensor(BLOCK_TIMESTEPS)
    out["time_sinusoid"] = proj(t).numpy()
    out["time_emb"] = emb(proj(t)).numpy()
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        if sys.argv[1] == "golden_fullsize_loop":             # python tools/make_golden.py golden_fullsize_loop [unipc:1.0 ...]
            golden_fullsize_loop(sys.argv[2:] or None)
            sys.exit(0)
        if sys.argv[1] == "golden_config_loops":              # python tools/make_golden.py golden_config_loops [c5 c3_r0 c3_r1]
            golden_config_loops(sys.argv[2:] or None)
            sys.exit(0)
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    check_full_schema()
    golden_pipeline_call()
    golden_lora_keys()
    golden_lora_merge()
    golden_blob_edit()
    golden_blob_edit_cv()
    golden_clip_text()
    golden_vae()
    golden_splat()
    golden_schedulers()
    golden_dinov2()
    golden_nets()
    golden_loop()
    golden_blocks()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-24s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))
