"""Drop-in module shells that keep the reference's nn.Module call signatures on top of the HIP plans.

  * UNet2DConditionModel.forward(sample, timestep, encoder_hidden_states, down_block_add_samples=[...],
        mid_block_add_sample=..., up_block_add_samples=[...], return_dict=False) -> (sample,)
        (D/models/unets/unet_2d_condition.py:1039-1057; the residual lists are consumed with pop(0) like :1217,1230,1313)
  * BlobNetModel.forward(sample, timestep, conditioning_scale: float, return_dict=False)
        -> (list[12], Tensor, list[15])            (blobctrl/models/blobnet.py:720-734, 941-945)
Tensors cross this boundary as NCHW torch tensors (fp32 or fp16) exactly like the reference; inside, everything is
NHWC fp16 and runs through libblobctrl_hip.  These shells are the module-level boundary; the captured-loop engine
(pipeline.py) bypasses them and chains the plans directly.
"""
from typing import List, Optional

import torch

from . import _lib
from .engine import Residuals, TrunkConfig, TrunkPlan
from .launch import Recorder, run_graphed
from .weights import PackedTrunk, pad8


def _stream():
    return torch.cuda.current_stream().cuda_stream


class ModelConfig(dict):
    """`module.config` with attribute AND mapping access, like diffusers' FrozenDict (pipe:957 reads `unet.config.in_channels`,
    pipe:993 `unet.config.time_cond_proj_dim`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class _TrunkModule(torch.nn.Module):
    """An `nn.Module` (DiffusionPipeline.register_modules, D/pipelines/pipeline_utils.py:788, accepts it; `.parameters()`, `.dtype`,
    `.device`, `.eval()`, `.to()` behave) whose forward runs a compiled launch plan.  The weights are the packed fp16 / fp32 arenas;
    they are exposed as two frozen Parameters that SHARE the arenas' storage.  `.to()` / `.half()` / `.float()` are no-ops: the layouts
    are fixed at construction (device = the one given to the constructor)."""

    def __init__(self, state_dict, config: TrunkConfig, device="cuda:0"):
        super().__init__()
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise _lib.BlobCtrlHipError("blobctrl_amd modules run on MI355X only; there is no CPU fallback")
        _lib.load()
        self.trunk_config = config
        latent = config.out_channels if config.out_channels else 4
        self.config = ModelConfig(
            # the reference script leaves unet.config.in_channels at the LATENT channel count (4) although conv_in has 5 inputs
            # (scripts/blobctrl_inference.py:233-249); BlobNet: in_channels=4, conditioning_channels=1+F (bn config)
            in_channels=latent if config.is_blobnet or config.in_channels == latent + 1 else config.in_channels,
            conv_in_channels=config.in_channels, out_channels=config.out_channels, block_out_channels=tuple(config.block_out_channels),
            layers_per_block=config.layers_per_block, attention_head_dim=config.num_heads, norm_num_groups=config.norm_num_groups,
            cross_attention_dim=config.cross_attention_dim, time_cond_proj_dim=None, sample_size=64,
            conditioning_channels=(config.in_channels - latent) if config.is_blobnet else None)
        self.weights = state_dict if isinstance(state_dict, PackedTrunk) else PackedTrunk(state_dict, self._device, config.block_out_channels)
        self.packed_fp16 = torch.nn.Parameter(self.weights.h_arena, requires_grad=False)
        self.packed_fp32 = torch.nn.Parameter(self.weights.f_arena, requires_grad=False)
        self._plans = {}

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return torch.float16

    def to(self, *a, **k):
        return self

    def _apply(self, fn, *a, **k):          # .cuda() / .half() / .float(): the packed layouts stay as built
        return self

    def _to_nhwc(self, rec, x: torch.Tensor, cpad: int) -> torch.Tensor:
        B, C, H, W = x.shape
        x = x.contiguous()
        if x.dtype not in (torch.float32, torch.float16):
            x = x.float()
        out = torch.empty(B, H * W, cpad, dtype=torch.float16, device=self.device)
        _lib.check(rec.lib.bc_nchw_to_nhwc_f16(x.data_ptr(), int(x.dtype == torch.float32), B, C, H * W, cpad,
                                               out.data_ptr(), _stream()), "bc_nchw_to_nhwc_f16")
        return out

    def _to_nchw(self, rec, t: torch.Tensor, B, C, H, W, dtype) -> torch.Tensor:
        out = torch.empty(B, C, H, W, dtype=dtype, device=self.device)
        _lib.check(rec.lib.bc_nhwc_to_nchw(t.data_ptr(), B, C, H * W, t.shape[-1], out.data_ptr(),
                                           int(dtype == torch.float32), _stream()), "bc_nhwc_to_nchw")
        return out


class BlobNetModel(_TrunkModule):
    def __init__(self, state_dict, config: TrunkConfig, device="cuda:0"):
        assert config.is_blobnet
        super().__init__(state_dict, config, device)

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", **_ignored):
        """`BlobNetModel.from_pretrained(blobnet_path, ignore_mismatched_sizes=True)` (inf:252): a directory with
        config.json + diffusion_pytorch_model.safetensors, or the .safetensors file itself."""
        from .checkpoint import load_blobnet
        sd, cfg = load_blobnet(path)
        return cls(sd, cfg, device)

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key not in self._plans:
            rec = Recorder(self.device)
            P = type("Plan", (), {})()
            P.rec = rec
            P.x_in = rec.zeros(B, H * W, pad8(self.trunk_config.in_channels))
            P.t = rec.zeros(1, dtype=torch.float32)
            P.idx = rec.zeros(1, dtype=torch.int32)
            P.scale = rec.zeros(1, dtype=torch.float32)
            P.seg = rec.begin("blobnet")
            plan = TrunkPlan(rec, self.weights, self.trunk_config, B, H, W)
            plan.record_time(P.t, P.idx)
            P.res = plan.record_forward(P.x_in, None, zero_scale=(1.0, P.scale, P.idx))
            P.shapes = plan.feat_shapes
            self._plans[key] = P
        return self._plans[key]

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, conditioning_scale: float = 1.0, return_dict: bool = False, **kw):
        if not isinstance(conditioning_scale, float):
            raise TypeError("conditioning_scale must be a Python float (pipeline_blobnet.py:395-396)")
        if return_dict:
            raise NotImplementedError("return_dict=True is broken in the reference (bn:947-956); use return_dict=False")
        B, C, H, W = sample.shape
        if C != self.trunk_config.in_channels:
            raise ValueError(f"expected {self.trunk_config.in_channels} input channels, got {C}")
        P = self._plan(B, H, W)
        P.x_in.copy_(self._to_nhwc(P.rec, sample, P.x_in.shape[-1]))
        P.t.fill_(float(timestep))
        P.scale.fill_(conditioning_scale)
        run_graphed(P.seg, self.device)
        dt = sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float32
        sd, sm, su = P.shapes
        down = [self._to_nchw(P.rec, r, B, c, h, w, dt) for r, (c, h, w) in zip(P.res.down, sd)]
        mid = self._to_nchw(P.rec, P.res.mid, B, sm[0], sm[1], sm[2], dt)
        up = [self._to_nchw(P.rec, r, B, c, h, w, dt) for r, (c, h, w) in zip(P.res.up, su)]
        return down, mid, up



class UNet2DConditionModel(_TrunkModule):
    def __init__(self, state_dict, config: TrunkConfig, device="cuda:0"):
        assert not config.is_blobnet
        super().__init__(state_dict, config, device)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, extra_in_channels=1, lora_path=None, lora_scale=1.0, device="cuda:0", **_ignored):
        """`UNet2DConditionModel.from_pretrained(sd15_path, subfolder="unet")` + the conv_in 4 -> 5 surgery (inf:229-249) +
        `load_lora_weights(unet_lora_path)` (inf:270-273) in one step: the packed weights are immutable, so the surgery and the
        LoRA merge happen before packing (`extra_in_channels=0` / `lora_path=None` skip them)."""
        import os
        from .checkpoint import load_unet
        sd, cfg = load_unet(os.path.join(path, subfolder) if subfolder else path, extra_in_channels, lora_path, lora_scale)
        return cls(sd, cfg, device)

    def _res_shapes(self, H, W):
        boc = self.trunk_config.block_out_channels
        nb = len(boc)
        down = [(boc[0], H, W)]
        h, w = H, W
        for i in range(nb):
            down += [(boc[i], h, w)] * self.trunk_config.layers_per_block
            if i < nb - 1:
                h, w = (h + 1) // 2, (w + 1) // 2
                down.append((boc[i], h, w))
        mid = (boc[-1], h, w)
        # up: mirror of the down resolutions
        sizes = []
        hh, ww = H, W
        for i in range(nb):
            sizes.append((hh, ww))
            if i < nb - 1:
                hh, ww = (hh + 1) // 2, (ww + 1) // 2
        rev = list(reversed(boc))
        up = []
        for i in range(nb):
            hh, ww = sizes[nb - 1 - i]
            up += [(rev[i], hh, ww)] * (self.trunk_config.layers_per_block + 1)
            if i < nb - 1:
                up.append((rev[i],) + sizes[nb - 2 - i])
        return down, mid, up

    def _plan(self, B, H, W, T, Dc, with_res):
        key = (B, H, W, T, Dc, with_res)
        if key not in self._plans:
            rec = Recorder(self.device)
            P = type("Plan", (), {})()
            P.rec = rec
            P.x_in = rec.zeros(B, H * W, pad8(self.trunk_config.in_channels))
            P.ctx = rec.zeros(B, T, Dc)
            P.t = rec.zeros(1, dtype=torch.float32)
            P.idx = rec.zeros(1, dtype=torch.int32)
            residuals = None
            if with_res:
                sd, sm, su = self._res_shapes(H, W)
                mk = lambda s: rec.zeros(B, s[1] * s[2], s[0])
                residuals = Residuals([mk(s) for s in sd], mk(sm), [mk(s) for s in su], bmod=B)
                P.res_shapes = (sd, sm, su)
            P.residuals = residuals
            P.seg = rec.begin("unet")
            plan = TrunkPlan(rec, self.weights, self.trunk_config, B, H, W)
            plan.record_context(P.ctx, T)
            plan.record_time(P.t, P.idx)
            P.eps = plan.record_forward(P.x_in, residuals)
            self._plans[key] = P
        return self._plans[key]

    def _fill_residual(self, P, buf: torch.Tensor, r: torch.Tensor, shape):
        """Place an NCHW residual (the right-hand square slice the pipeline passes, pipe:1085-1087) into the canvas."""
        C, H, W = shape
        B = r.shape[0]
        ws = r.shape[-1]
        tmp = self._to_nhwc(P.rec, r, C)
        buf.view(B, H, W, C)[:, :, W - ws:, :].copy_(tmp.view(B, H, ws, C))

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, timestep_cond=None, cross_attention_kwargs=None,
                down_block_add_samples: Optional[List[torch.Tensor]] = None, mid_block_add_sample=None,
                up_block_add_samples: Optional[List[torch.Tensor]] = None, added_cond_kwargs=None,
                return_dict: bool = False, **kw):
        B, C, H, W = sample.shape
        if C != self.trunk_config.in_channels:
            raise ValueError(f"expected {self.trunk_config.in_channels} input channels, got {C}")
        is_blobnet = (down_block_add_samples is not None and mid_block_add_sample is not None
                      and up_block_add_samples is not None)                        # unet_2d_condition.py:1200
        T, Dc = encoder_hidden_states.shape[1:]
        P = self._plan(B, H, W, T, Dc, is_blobnet)
        P.x_in.copy_(self._to_nhwc(P.rec, sample, P.x_in.shape[-1]))
        P.ctx.copy_(encoder_hidden_states.to(self.device, torch.float16))
        P.t.fill_(float(timestep))
        if is_blobnet:
            sd, sm, su = P.res_shapes
            if len(down_block_add_samples) != len(sd) or len(up_block_add_samples) != len(su):
                raise ValueError("wrong number of BlobNet residuals")
            for buf, s in zip(P.residuals.down, sd):
                self._fill_residual(P, buf, down_block_add_samples.pop(0), s)     # lists are consumed (:1217,1230)
            self._fill_residual(P, P.residuals.mid, mid_block_add_sample, sm)
            for buf, s in zip(P.residuals.up, su):
                self._fill_residual(P, buf, up_block_add_samples.pop(0), s)        # (:1313)
        run_graphed(P.seg, self.device)
        dt = sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float32
        out = torch.empty(B, self.trunk_config.out_channels, H, W, dtype=torch.float32, device=self.device)
        out.copy_(P.eps.view(B, H, W, self.trunk_config.out_channels).permute(0, 3, 1, 2))
        return (out.to(dt),)

