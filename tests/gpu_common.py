"""Helpers shared by the `-m gpu` tests (everything here goes through the C ABI of libblobctrl_hip.so)."""
import torch

from blobctrl_amd.engine import TrunkConfig
from tests.common import TINY


def tiny_trunk_configs():
    c = TINY
    u = TrunkConfig(in_channels=5, block_out_channels=c["boc"], num_heads=c["heads"], norm_num_groups=c["groups"],
                    cross_attention_dim=c["ctx"], out_channels=4, is_blobnet=False)
    b = TrunkConfig(in_channels=4 + 1 + c["feat"], block_out_channels=c["boc"], num_heads=c["heads"],
                    norm_num_groups=c["groups"], cross_attention_dim=None, out_channels=0, is_blobnet=True)
    return u, b


def make_pipeline(usd, bsd, scheduler="unipc", use_graphs=True):
    from blobctrl_amd.pipeline import BlobCtrlEngine
    u, b = tiny_trunk_configs()
    return BlobCtrlEngine(usd, bsd, u, b, device="cuda:0", scheduler=scheduler, use_graphs=use_graphs)
