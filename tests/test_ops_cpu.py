"""`torch.ops.blobctrl.*` (blobctrl_amd/ops.py): the hot path's four calls registered with torch.library over the C ABI.  Without a GPU:
the ops exist with the expected schemas, their fake (meta) kernels state the output shapes / dtypes / devices of the reference calls
(blobnet.py:720-945 residual lists, unet_2d_condition.py:1039-1353 eps, utils.py:145-194 scores, pipeline_blobnet.py:1025-1123 latents)
under FakeTensorMode, and the module shells route through them.  The real kernels need an MI355X (tests/test_ops_gpu.py)."""
import pytest
import torch

from tests.common import TINY, tiny_weights
from tests.gpu_common import tiny_trunk_configs


def _modules(device="cpu"):
    from blobctrl_amd.modules import BlobNetModel, UNet2DConditionModel
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_trunk_configs()
    return UNet2DConditionModel(usd, ucfg, device, lazy=True), BlobNetModel(bsd, bcfg, device, lazy=True), ucfg, bcfg


def test_ops_are_registered_with_the_expected_schemas():
    from blobctrl_amd import ops  # noqa: F401
    schemas = {n: str(getattr(torch.ops.blobctrl, n).default._schema) for n in ("splat_scores", "blobnet_forward", "unet_forward", "denoise")}
    assert schemas["splat_scores"] == "blobctrl::splat_scores(Tensor params, SymInt h, SymInt w, SymInt device_index) -> Tensor"
    assert schemas["blobnet_forward"].endswith("-> Tensor[]") and "float conditioning_scale" in schemas["blobnet_forward"]
    assert "Tensor[] down_add, Tensor? mid_add, Tensor[] up_add" in schemas["unet_forward"]
    assert "float[] conditioning_scales" in schemas["denoise"] and schemas["denoise"].endswith("-> Tensor")


def test_fake_kernels_state_the_reference_output_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    from blobctrl_amd import ops
    unet, blobnet, ucfg, bcfg = _modules()
    B, H, W = 2, 8, 16
    with FakeTensorMode():
        sc = torch.ops.blobctrl.splat_scores(torch.zeros(1, 8, dtype=torch.float64), 8, 12, 0)
        assert tuple(sc.shape) == (1, 2, 8, 12) and sc.dtype == torch.float64 and sc.device.type == "cuda"
        outs = torch.ops.blobctrl.blobnet_forward(torch.empty(B, bcfg.in_channels, H, W), 981.0, 1.0, ops.register(blobnet))
        down, mid, up = ops.blobnet_output_shapes(bcfg, B, H, W)
        assert [tuple(o.shape[1:]) for o in outs] == down + [mid] + up and len(down) == 12 and len(up) == 15           # the 28 residuals of bn:860-924
        assert outs[0].shape[0] == B and outs[-1].dtype == torch.float32
        eps = torch.ops.blobctrl.unet_forward(torch.empty(B, ucfg.in_channels, H, W), 981.0, torch.empty(B, 7, TINY["ctx"]), list(outs[:12]), outs[12],
                                              list(outs[13:]), ops.register(unet))
        assert tuple(eps.shape) == (B, 4, H, W) and eps.dtype == torch.float32
    # the shapes the fake kernel states are the ones the UNet shell expects for its residual buffers
    assert (down, mid, up) == tuple(unet._res_shapes(H, W))


def test_real_kernels_refuse_to_run_without_the_gpu():
    from blobctrl_amd import _lib, ops
    from blobctrl_amd.splat import splat_features
    with pytest.raises(_lib.BlobCtrlHipError):
        splat_features(torch.tensor([0.5]), torch.tensor([0.5]), torch.eye(2).reshape(1, 1, 2, 2) * 0.01, torch.tensor([[1.0]]), score_size=(8, 8),
                       return_d_score=True, device="cpu")
    with pytest.raises(RuntimeError):
        torch.ops.blobctrl.unet_forward(torch.zeros(1, 5, 8, 8), 1.0, torch.zeros(1, 7, TINY["ctx"]), [], None, [], 10 ** 9)     # dead handle
