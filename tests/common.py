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
