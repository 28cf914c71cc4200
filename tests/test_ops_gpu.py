"""`torch.ops.blobctrl.*` on the MI355X: torch.library.opcheck (schema + fake-tensor consistency against the REAL kernels) and the
shells' route through the dispatcher (a TorchDispatchMode sees the ops; results equal the direct `_forward_impl` calls)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.common import TINY, g, tiny_weights  # noqa: E402
from tests.gpu_common import tiny_trunk_configs  # noqa: E402


def test_opcheck_and_dispatch_of_the_module_ops():
    from torch.utils._python_dispatch import TorchDispatchMode
    from blobctrl_amd import ops
    from blobctrl_amd.modules import BlobNetModel, UNet2DConditionModel
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_trunk_configs()
    unet, blobnet = UNet2DConditionModel(usd, ucfg, "cuda:0"), BlobNetModel(bsd, bcfg, "cuda:0")
    B, H, W = 2, 8, 16
    xb, xu, ctx = g(1, B, bcfg.in_channels, H, W).cuda(), g(2, B, ucfg.in_channels, H, W).cuda(), g(3, B, 7, TINY["ctx"]).cuda()
    hb, hu = ops.register(blobnet), ops.register(unet)
    checks = ("test_schema", "test_faketensor")
    sq = lambda t: t[..., t.shape[-1] - t.shape[-2]:].contiguous()          # the right-hand square the pipeline passes on (pipe:1085-1087)
    torch.library.opcheck(torch.ops.blobctrl.blobnet_forward, (xb, 981.0, 0.8, hb), test_utils=checks)
    outs = torch.ops.blobctrl.blobnet_forward(xb, 981.0, 0.8, hb)
    torch.library.opcheck(torch.ops.blobctrl.unet_forward, (xu, 981.0, ctx, [sq(o) for o in outs[:12]], sq(outs[12]), [sq(o) for o in outs[13:]], hu),
                          test_utils=checks)
    torch.library.opcheck(torch.ops.blobctrl.splat_scores, (torch.tensor([[0.4, 0.6, 0.01, 0.002, 0.002, 0.02, 1.0, 0.0]], dtype=torch.float64), 8, 16, 0),
                          test_utils=checks)

    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if "blobctrl" in str(func):
                seen.append(str(func))
            return func(*args, **(kwargs or {}))
    with Spy():
        down, mid, up = blobnet(xb, 981.0, conditioning_scale=0.8)
        dl, ul = [sq(d) for d in down], [sq(u) for u in up]
        eps = unet(xu, 981.0, ctx, down_block_add_samples=dl, mid_block_add_sample=sq(mid), up_block_add_samples=ul)[0]
    assert seen == ["blobctrl.blobnet_forward.default", "blobctrl.unet_forward.default"] and dl == [] and ul == []      # lists consumed
    ref = unet._forward_impl(xu, 981.0, ctx, [sq(d) for d in down], sq(mid), [sq(u) for u in up])
    assert torch.equal(eps, ref)
    assert all(torch.equal(a, b) for a, b in zip(list(down) + [mid] + list(up), outs))


def test_engine_call_goes_through_the_denoise_op():
    from torch.utils._python_dispatch import TorchDispatchMode
    from tests.gpu_common import make_pipeline
    from oracle import blob_splat
    usd, bsd = tiny_weights()
    pipe = make_pipeline(usd, bsd, scheduler="ddim")
    score = torch.from_numpy(blob_splat.splat_scores_from_ellipse([[40.0, 42.0], [20.0, 30.0], 25.0], 64, 64, 8, 8))
    args = (g(32, 2, 7, TINY["ctx"]), g(33, 1, 4, 8, 8), g(34, 1, 4, 8, 8), score, g(35, 1, 1, TINY["feat"]))
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if "blobctrl" in str(func):
                seen.append(str(func))
            return func(*args, **(kwargs or {}))
    with Spy():
        a = pipe(*args, num_inference_steps=3, guidance_scale=7.5, latents=g(31, 1, 4, 8, 8))
    b = pipe.denoise(*args, num_inference_steps=3, guidance_scale=7.5, latents=g(31, 1, 4, 8, 8))
    assert seen == ["blobctrl.denoise.default"] and torch.equal(a, b)


def test_list_valued_conditioning_scale_is_validated_identically_through_the_op():
    """`blobnet_conditioning_scale=[s]` for a per-request batch of 2 is a list of the wrong length - the dispatcher path
    used to collapse it to a scalar and broadcast it; it must raise exactly what the direct `denoise` call raises.  A correct
    per-request list, and a one-element list for a single edit, give the direct call's result bit for bit."""
    from tests.gpu_common import make_pipeline
    from oracle import blob_splat
    usd, bsd = tiny_weights()
    pipe = make_pipeline(usd, bsd, scheduler="ddim")
    sc = torch.from_numpy(blob_splat.splat_scores_from_ellipse([[40.0, 42.0], [20.0, 30.0], 25.0], 64, 64, 8, 8))
    two = (g(32, 4, 7, TINY["ctx"]), g(33, 2, 4, 8, 8), g(34, 2, 4, 8, 8), torch.cat([sc, sc]), g(35, 2, 1, TINY["feat"]))
    kw = dict(num_inference_steps=2, guidance_scale=7.5, latents=g(31, 2, 4, 8, 8))
    for call in (pipe, pipe.denoise):
        with pytest.raises(ValueError, match="expected 2 values, got 1"):
            call(*two, blobnet_conditioning_scale=[0.7], **kw)
    a = pipe(*two, blobnet_conditioning_scale=[0.7, 0.0], **kw)
    b = pipe.denoise(*two, blobnet_conditioning_scale=[0.7, 0.0], **kw)
    assert torch.equal(a, b)
    one = (g(32, 2, 7, TINY["ctx"]), g(33, 1, 4, 8, 8), g(34, 1, 4, 8, 8), sc, g(35, 1, 1, TINY["feat"]))
    kw1 = dict(num_inference_steps=2, guidance_scale=7.5, latents=g(31, 1, 4, 8, 8))
    assert torch.equal(pipe(*one, blobnet_conditioning_scale=[0.7], **kw1), pipe.denoise(*one, blobnet_conditioning_scale=0.7, **kw1))
