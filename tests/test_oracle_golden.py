"""Pins the oracle (oracle/*) against fixtures produced by the REAL reference (tools/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import blob_splat, dinov2 as o_dino, nets, pipeline as o_pipe, schedulers as o_sched
from blobctrl_amd import synth
from tests.common import TINY, g, tiny_cfgs, tiny_weights

torch.set_grad_enabled(False)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_splat_known_answer_and_fixtures(golden_dir):
    z = load(golden_dir, "splat.npz")
    meta = json.loads(str(z["meta"]))
    for i, m in enumerate(meta):
        mean, cov = blob_splat.ellipse_to_gaussian(m["ellipse"])
        nm, nc = blob_splat.normalize_gaussian(mean, cov, m["W"], m["H"])
        np.testing.assert_allclose(nm, z[f"mean_{i}"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(nc, z[f"cov_{i}"], rtol=1e-13, atol=1e-18)
        s = blob_splat.splat_scores(nm[0], nm[1], nc, 1.0, m["h"], m["w"])
        ref = z[f"score_{i}"]
        assert s.shape == ref.shape and s.dtype == np.float64
        np.testing.assert_allclose(s, ref, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(s[0, 0] + s[0, 1], 1.0, rtol=0, atol=1e-15)       # bg + fg == 1
        if m["name"] == "move_hat_cli":                                                # SURVEY 8c known answer
            assert abs(s[0, 1].sum() - 75.369095) < 1e-5
            assert abs(s[0, 1].max() - 0.999534) < 1e-6
            assert np.unravel_index(s[0, 1].argmax(), (64, 64)) == (46, 45)
            assert abs(s[0, 1, 45, 45] - 0.965685) < 1e-6
    s = blob_splat.splat_scores(z["mean_0"][0], z["mean_0"][1], z["cov_0"], 0.2, 64, 64)   # size<0.5 branch
    np.testing.assert_allclose(s, z["score_absent"], rtol=1e-12)


@pytest.mark.parametrize("n", [5, 20, 50])
def test_schedulers(golden_dir, n):
    z = load(golden_dir, "schedulers.npz")
    for name, cls in (("unipc", o_sched.UniPCOracle), ("ddim", o_sched.DDIMOracle)):
        sch = cls()
        sch.set_timesteps(n)
        np.testing.assert_array_equal(sch.timesteps.numpy(), z[f"{name}_{n}_timesteps"])
        traj = z[f"{name}_{n}_traj"]
        x = torch.from_numpy(traj[0])
        for i in range(n):
            x = sch.step(g(100 + i, 1, 4, 8, 8), x)
            np.testing.assert_allclose(x.numpy(), traj[i + 1], rtol=2e-5, atol=2e-5 * np.abs(traj[i + 1]).max())
    np.testing.assert_array_equal(o_sched.UniPCOracle().__class__().alphas_cumprod.numpy().shape, (1000,))
    u = o_sched.UniPCOracle()
    u.set_timesteps(n)
    np.testing.assert_allclose(u.sigmas.numpy(), z[f"unipc_{n}_sigmas"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["wide", "square", "odd"])
def test_nets_tiny(golden_dir, tag):
    z = load(golden_dir, "nets_tiny.npz")
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_cfgs()
    t = torch.tensor(int(z["timestep"]))
    down, mid, up = nets.blobnet_forward(bsd, bcfg, torch.from_numpy(z[f"{tag}_blob_in"]), t, 0.8)
    assert len(down) == 12 and len(up) == 15
    for i, r in enumerate(down):
        np.testing.assert_allclose(r.numpy(), z[f"{tag}_down_{i}"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(mid.numpy(), z[f"{tag}_mid"], rtol=1e-4, atol=2e-5)
    for i, r in enumerate(up):
        np.testing.assert_allclose(r.numpy(), z[f"{tag}_up_{i}"], rtol=1e-4, atol=2e-5)
    sq = lambda r: r[..., -r.shape[-2]:]
    x = torch.from_numpy(z[f"{tag}_unet_in"])
    ehs = torch.from_numpy(z[f"{tag}_ehs"])
    eps = nets.unet_forward(usd, ucfg, x, t, ehs, [sq(r) for r in down], sq(mid), [sq(r) for r in up])
    np.testing.assert_allclose(eps.numpy(), z[f"{tag}_eps"], rtol=1e-4, atol=5e-5)
    eps_plain = nets.unet_forward(usd, ucfg, x, t, ehs)
    np.testing.assert_allclose(eps_plain.numpy(), z[f"{tag}_eps_plain"], rtol=1e-4, atol=5e-5)
    assert np.abs(z[f"{tag}_eps"] - z[f"{tag}_eps_plain"]).max() > 0.1          # the coupling is live


@pytest.mark.parametrize("tag", ["unipc_5", "unipc_6", "ddim_5", "ddim_6"])
def test_loop_tiny(golden_dir, tag):
    z = load(golden_dir, "loop_tiny.npz")
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_cfgs()
    sname, steps = tag.split("_")
    sch = o_sched.UniPCOracle() if sname == "unipc" else o_sched.DDIMOracle()
    gs, ge = z[f"{tag}_window"]
    score = blob_splat.splat_scores_from_ellipse([[40.0, 42.0], [20.0, 30.0], 25.0], 64, 64, 8, 8)
    np.testing.assert_allclose(score, z["gs_score"], rtol=1e-9)
    trace = []
    final = o_pipe.denoise_loop(usd, ucfg, bsd, bcfg, sch, int(steps), g(31, 1, 4, 8, 8), g(32, 2, 7, TINY["ctx"]),
                                g(33, 1, 4, 8, 8) * 0.18215 * 5, g(34, 1, 4, 8, 8) * 0.18215 * 5,
                                torch.from_numpy(score), g(35, 1, 1, TINY["feat"]), 7.5, 1.0, float(gs), float(ge),
                                trace=trace)
    eps_ref = z[f"{tag}_eps"]
    np.testing.assert_allclose(trace[0][1].numpy(), eps_ref[0], rtol=1e-4, atol=1e-4)
    ref = z[f"{tag}_final"]
    err = np.abs(final.numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-3, err          # free-running fp32 on non-contractive random weights: reorder noise grows


def test_dinov2_oracle_vs_fixture(golden_dir):
    z = load(golden_dir, "dinov2_tiny.npz")
    sd = synth.synth_state_dict(synth.dinov2_param_shapes(64, 3, 4, 14, 25), 99)
    for tag in ("native", "interp"):
        out = o_dino.dinov2_pooled(sd, torch.from_numpy(z[f"{tag}_in"]), 4, 14)
        np.testing.assert_allclose(out.numpy(), z[f"{tag}_pooled"], rtol=1e-4, atol=1e-5)


def test_dinov2_oracle_vs_transformers():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.Dinov2Config(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, image_size=42,
                                    patch_size=14, mlp_ratio=4)
    model = transformers.Dinov2Model(cfg).eval()
    sd = synth.synth_state_dict(synth.dinov2_param_shapes(32, 2, 4, 14, 9), 5)
    model.load_state_dict(sd, strict=True)
    x = g(3, 1, 3, 42, 42)
    np.testing.assert_allclose(o_dino.dinov2_pooled(sd, x, 2, 14).numpy(), model(pixel_values=x).pooler_output.numpy(),
                               rtol=1e-4, atol=1e-5)


def test_vae_oracle_vs_reference_fixture(golden_dir):
    from oracle import vae as o_vae
    z = load(golden_dir, "vae_tiny.npz")
    sd = synth.synth_state_dict(synth.vae_param_shapes((32, 32, 64, 64), 2, 4), 21)
    mom = o_vae.encode_moments(sd, torch.from_numpy(z["img"]), groups=8)
    np.testing.assert_allclose(mom.numpy(), z["moments"], rtol=1e-4, atol=1e-5)
    smp = o_vae.sample_posterior(mom, torch.from_numpy(z["noise"]), scale=1.0)
    np.testing.assert_allclose(smp.numpy(), z["sample"], rtol=1e-4, atol=1e-5)
    dec = o_vae.decode(sd, torch.from_numpy(z["z"]), groups=8)
    np.testing.assert_allclose(dec.numpy(), z["decoded"], rtol=1e-4, atol=1e-5)


def test_clip_text_oracle_vs_transformers_fixture(golden_dir):
    from oracle.clip_text import clip_text_hidden
    z = load(golden_dir, "clip_text_tiny.npz")
    sd = synth.synth_state_dict(synth.clip_text_param_shapes(99, 64, 3, 128, 77), 77)
    ids = torch.from_numpy(z["ids"])
    np.testing.assert_allclose(clip_text_hidden(sd, ids, 4).numpy(), z["last_hidden_state"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(clip_text_hidden(sd, ids, 4, clip_skip=1).numpy(), z["clip_skip_1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(clip_text_hidden(sd, torch.from_numpy(z["short_ids"]), 4).numpy(), z["short_last_hidden_state"],
                               rtol=0, atol=2e-5)
    # both key layouts (with / without the "text_model." prefix) are accepted
    bare = {k[len("text_model."):]: v for k, v in sd.items()}
    assert torch.equal(clip_text_hidden(bare, ids, 4), clip_text_hidden(sd, ids, 4))


def oracle_block(name):
    """The oracle's restatement of one block case of tests/common.py BLOCK_CASES."""
    from tests.common import BLOCK_CASES, block_inputs, block_weights
    kind, p = BLOCK_CASES[name]
    sd = block_weights(name)
    x, temb, ctx = block_inputs(name)
    if kind == "resnet":
        return nets.resnet_block(sd, "", x, temb, 32)
    if kind == "transformer":
        return nets.transformer_2d(sd, "", x, ctx, p["heads"], 32)
    if kind == "upsample":
        return nets.upsample(sd, "", x, size=p["size"])
    return nets.downsample(sd, "", x)


@pytest.mark.parametrize("name", ["res_320_320", "res_320_640_shortcut", "res_2560_1280", "tfm_320_cross", "tfm_320_self_only",
                                  "up_scale2", "up_explicit_size", "down", "tfm_640_self_only", "tfm_1280_cross"])
def test_blocks_vs_reference_fixture(golden_dir, name):
    """SURVEY 8c.2 block-level goldens (diffusers' own ResnetBlock2D / Transformer2DModel / Upsample2D / Downsample2D at SD-1.5 widths;
    the 640- / 1280-channel transformer cases keep every `sub`-th pixel of the reference output)."""
    from tests.common import BLOCK_CASES
    z = load(golden_dir, "blocks.npz")
    sub = BLOCK_CASES[name][1].get("sub", 1)
    y = oracle_block(name).numpy()[:, :, ::sub, ::sub]
    assert y.shape == z[name].shape
    np.testing.assert_allclose(y, z[name], rtol=1e-4, atol=2e-4 if sub > 1 else 1e-4)


def test_time_embedding_vs_reference_fixture(golden_dir):
    from tests.common import BLOCK_TIMESTEPS, block_weights
    z = load(golden_dir, "blocks.npz")
    t = torch.tensor(BLOCK_TIMESTEPS)
    np.testing.assert_allclose(nets.timestep_embedding(t, 320).numpy(), z["time_sinusoid"], rtol=1e-5, atol=1e-5)
    sd = {"time_embedding." + k: v for k, v in block_weights("time").items()}
    cfg = nets.NetConfig(in_channels=4)
    for i, ti in enumerate(BLOCK_TIMESTEPS):
        np.testing.assert_allclose(nets.time_embed(sd, ti, 1, cfg).numpy()[0], z["time_emb"][i], rtol=1e-4, atol=1e-4)
