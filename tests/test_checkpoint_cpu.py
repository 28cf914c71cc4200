"""Weight / LoRA converter: safetensors reader, conv_in surgery (inf:233-249), LoRA key renaming and alpha
defaults pinned against the reference's own functions (tests/golden/lora_keys.json, made by tools/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from blobctrl_amd import checkpoint as ck
from blobctrl_amd import synth
from blobctrl_amd.weights import merge_lora
from tests.common import TINY, g


def test_safetensors_round_trip_and_library_compat(tmp_path):
    t = {"a.weight": g(1, 3, 5), "b": g(2, 7).half(), "n": torch.arange(6).reshape(2, 3), "e": torch.zeros(0, 4)}
    p = str(tmp_path / "x.safetensors")
    ck.write_safetensors(p, t, {"format": "pt"})
    back = ck.read_safetensors(p)
    assert list(back) == list(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and torch.equal(back[k], t[k])
    st = pytest.importorskip("safetensors.torch")                   # the published implementation reads our file and vice versa
    lib = st.load_file(p)
    assert all(torch.equal(lib[k], t[k]) for k in t)
    p2 = str(tmp_path / "y.safetensors")
    st.save_file({"w": g(3, 4, 4), "h": g(4, 2, 2).to(torch.bfloat16)}, p2)
    mine = ck.read_safetensors(p2)
    assert torch.equal(mine["w"], g(3, 4, 4)) and torch.equal(mine["h"], g(4, 2, 2).to(torch.bfloat16).float())
    with open(str(tmp_path / "bad.safetensors"), "wb") as f:
        f.write(b"\x01\x02")
    with pytest.raises(ValueError):
        ck.read_safetensors(str(tmp_path / "bad.safetensors"))


def test_lora_key_conversion_matches_reference(golden_dir):
    z = json.load(open(os.path.join(golden_dir, "lora_keys.json")))
    for src, want in z["pairs"]:
        assert ck.convert_unet_lora_key(src) == want


def _write_lora(path, mods, ranks, alphas=None, fmt="peft", prefix="unet."):
    t = {}
    for i, (m, shape) in enumerate(mods.items()):
        r = ranks[i]
        a = g(100 + i, r, *shape[1:]) * 0.1
        b = g(200 + i, shape[0], r, *([1] * (len(shape) - 2))) * 0.1
        if fmt == "peft":
            t[f"{prefix}{m}.lora_A.weight"], t[f"{prefix}{m}.lora_B.weight"] = a, b
        else:
            t[f"{prefix}{m}.lora.down.weight"], t[f"{prefix}{m}.lora.up.weight"] = a, b
        if alphas:
            t[f"{prefix}{m}.alpha"] = torch.tensor(float(alphas[i]))
    ck.write_safetensors(path, t)
    return t


def test_lora_alpha_defaults_follow_peft_kwargs(tmp_path, golden_dir):
    """No alpha keys: alpha = FIRST rank for every module (peft_utils.py:153); alpha keys: per-module alpha."""
    z = json.load(open(os.path.join(golden_dir, "lora_keys.json")))
    mods = {"a": (6, 5), "b": (6, 5), "c": (4, 5, 3, 3)}
    p = str(tmp_path / "l.safetensors")
    _write_lora(p, mods, [8, 4, 4])
    lora, alphas = ck.load_lora(p)
    kw = z["kwargs_no_alpha"]
    assert alphas == {"a": kw["lora_alpha"], "b": kw["lora_alpha"], "c": kw["lora_alpha"]} and kw["lora_alpha"] == 8
    _write_lora(p, mods, [8, 4, 4], alphas=[16.0, 4.0, 4.0], fmt="old")
    lora, alphas = ck.load_lora(p)
    kw = z["kwargs_alpha"]
    assert alphas == {"a": kw["alpha_pattern"]["a"], "b": kw["lora_alpha"], "c": kw["lora_alpha"]}
    # merged weight = W + (alpha / r) B A, Linear and Conv2d
    sd = {"a.weight": g(1, 6, 5), "b.weight": g(2, 6, 5), "c.weight": g(3, 4, 5, 3, 3)}
    merged = merge_lora(sd, lora, alphas)
    for m, r in (("a", 8), ("b", 4), ("c", 4)):
        A, B = lora[m + ".lora_A.weight"], lora[m + ".lora_B.weight"]
        want = sd[m + ".weight"] + (alphas[m] / r) * (B.flatten(1) @ A.flatten(1)).reshape(sd[m + ".weight"].shape)
        torch.testing.assert_close(merged[m + ".weight"], want)
    # a Conv2d LoRA is the composition of the two convolutions
    x = g(9, 1, 5, 6, 6)
    A, B = lora["c.lora_A.weight"], lora["c.lora_B.weight"]
    two = torch.nn.functional.conv2d(torch.nn.functional.conv2d(x, A, padding=1), B)
    one = torch.nn.functional.conv2d(x, merged["c.weight"] - sd["c.weight"], padding=1) / (alphas["c"] / 4)
    torch.testing.assert_close(one, two, rtol=1e-4, atol=1e-5)


def test_load_unet_blobnet_from_disk_with_surgery_and_lora(tmp_path):
    c = TINY
    base = synth.synth_state_dict(synth.trunk_param_shapes(4, c["boc"], 2, c["ctx"], 4, blobnet=False), 5)
    ud = tmp_path / "sd15" / "unet"
    ud.mkdir(parents=True)
    ck.write_safetensors(str(ud / "diffusion_pytorch_model.safetensors"), {k: v.half() for k, v in base.items()})
    json.dump({"block_out_channels": list(c["boc"]), "attention_head_dim": c["heads"], "norm_num_groups": c["groups"],
               "cross_attention_dim": c["ctx"], "in_channels": 4}, open(ud / "config.json", "w"))
    sd, cfg = ck.load_unet(str(ud))
    assert cfg.in_channels == 5 and cfg.num_heads == c["heads"] and cfg.cross_attention_dim == c["ctx"] and not cfg.is_blobnet
    w = sd["conv_in.weight"]
    assert w.shape[1] == 5 and torch.equal(w[:, :4], base["conv_in.weight"].half().float()) and float(w[:, 4].abs().max()) == 0.0
    assert torch.equal(sd["conv_in.bias"], base["conv_in.bias"].half().float())
    # LoRA over the 5-channel conv_in, an attention projection and a ResBlock conv
    mods = {"conv_in": tuple(w.shape), "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q":
            tuple(sd["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"].shape),
            "mid_block.resnets.0.conv1": tuple(sd["mid_block.resnets.0.conv1.weight"].shape)}
    lp = tmp_path / "unet_lora"
    lp.mkdir()
    _write_lora(str(lp / "pytorch_lora_weights.safetensors"), mods, [4, 4, 4])
    sd2, _ = ck.load_unet(str(ud), lora_path=str(lp))
    changed = [k for k in sd if not torch.equal(sd[k], sd2[k])]
    assert sorted(changed) == sorted(m + ".weight" for m in mods)
    assert float(sd2["conv_in.weight"][:, 4].abs().max()) > 0          # the adapter writes the new channel
    _write_lora(str(lp / "pytorch_lora_weights.safetensors"), {"nope.layer": (4, 4)}, [2])
    with pytest.raises(KeyError):
        ck.load_unet(str(ud), lora_path=str(lp))
    # BlobNet
    bsd = synth.synth_state_dict(synth.trunk_param_shapes(4 + 1 + c["feat"], c["boc"], 2, None, None, blobnet=True), 6)
    bd = tmp_path / "blobnet"
    bd.mkdir()
    ck.write_safetensors(str(bd / "diffusion_pytorch_model.safetensors"), bsd)
    json.dump({"block_out_channels": list(c["boc"]), "attention_head_dim": c["heads"], "norm_num_groups": c["groups"],
               "in_channels": 4, "conditioning_channels": 1 + c["feat"]}, open(bd / "config.json", "w"))
    sdb, cfgb = ck.load_blobnet(str(bd))
    assert cfgb.is_blobnet and cfgb.in_channels == 5 + c["feat"] and cfgb.cross_attention_dim is None
    assert all(torch.equal(sdb[k], bsd[k]) for k in bsd)
    json.dump({"in_channels": 4, "conditioning_channels": 3}, open(bd / "config.json", "w"))
    with pytest.raises(ValueError):
        ck.load_blobnet(str(bd))


@pytest.mark.parametrize("kind", ["lin", "conv"])
@pytest.mark.parametrize("tag", ["noalpha", "alpha"])
def test_lora_merge_matches_reference_fuse_and_runtime_branch(golden_dir, kind, tag):
    """f2 pinned: `weights.merge_lora` against the adapter arithmetic the reference tree holds itself -
    D/models/lora.py LoRALinearLayer / LoRAConv2dLayer forward (base(x) + lora_scale * up(down(x)) [* network_alpha / rank],
    :175-233, :235-297) and LoRACompatible*._fuse_lora (:317-340, :398-424); fixture tests/golden/lora_merge.npz from those classes
    (tools/make_golden.py golden_lora_merge), Linear and Conv2d 3x3, with and without network_alpha, lora_scale 0.7."""
    z = np.load(os.path.join(golden_dir, "lora_merge.npz"))
    t = lambda n: torch.from_numpy(z[f"{kind}_{tag}_{n}"])
    rank = t("down").shape[0]
    # peft naming: lora_A = down, lora_B = up (D/utils/state_dict_utils.py UNET_TO_PEFT); alpha defaults to the rank (no scaling)
    alpha = float(z["network_alpha"]) if tag == "alpha" else float(rank)
    merged = merge_lora({"m.weight": t("w"), "m.bias": t("b")}, {"m.lora_A.weight": t("down"), "m.lora_B.weight": t("up")},
                        {"m": alpha}, adapter_scale=float(z["lora_scale"]))
    assert torch.equal(merged["m.bias"], t("b"))
    assert (merged["m.weight"] - t("w_fused")).abs().max().item() < 1e-6
    x = t("x")
    y = torch.nn.functional.linear(x, merged["m.weight"], merged["m.bias"]) if kind == "lin" else \
        torch.nn.functional.conv2d(x, merged["m.weight"], merged["m.bias"], padding=1)
    assert (y - t("y_fused")).abs().max().item() < 1e-5
    assert (y - t("y_runtime")).abs().max().item() < 1e-5          # fused == the runtime low-rank branch, as the reference executes it
