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
from collections import OrderedDict
from typing import List, Optional

import torch

from . import _lib
from .engine import Residuals, TrunkConfig, TrunkPlan
from .launch import Recorder, run_graphed
from .weights import PackedTrunk, merge_lora, pad8


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

    def __init__(self, state_dict, config: TrunkConfig, device="cuda:0", lazy: bool = False):
        """`lazy=True` (what `from_pretrained` uses): the reference-schema state dict stays on the host and is packed on first
        use, so that the script's post-load edits - the conv_in surgery (inf:233-249), `load_lora_weights` (inf:270-273) - are
        cheap host operations; any later edit drops the packed copy and every cached plan (`_invalidate`)."""
        super().__init__()
        self._device = torch.device(device)
        self.trunk_config = config
        self._sd = None if isinstance(state_dict, PackedTrunk) else OrderedDict(state_dict)
        self._packed = state_dict if isinstance(state_dict, PackedTrunk) else None
        self._adapters = OrderedDict()          # name -> dict(lora=, alphas=, weight=, active=)
        self._version = 0
        self._plans = {}
        self._make_config()
        if not lazy and self._packed is None:
            if self._device.type != "cuda":
                raise _lib.BlobCtrlHipError("blobctrl_amd modules run on MI355X only; there is no CPU fallback")
            self.weights                         # pack now

    def _make_config(self):
        config = self.trunk_config
        latent = config.out_channels if config.out_channels else 4
        self.config = ModelConfig(
            # the reference script leaves unet.config.in_channels at the LATENT channel count (4) although conv_in has 5 inputs
            # (scripts/blobctrl_inference.py:233-249); BlobNet: in_channels=4, conditioning_channels=1+F (bn config)
            in_channels=latent if config.is_blobnet or config.in_channels in (latent, latent + 1) else config.in_channels,
            conv_in_channels=config.in_channels, out_channels=config.out_channels, block_out_channels=tuple(config.block_out_channels),
            layers_per_block=config.layers_per_block, attention_head_dim=config.num_heads, norm_num_groups=config.norm_num_groups,
            cross_attention_dim=config.cross_attention_dim, time_cond_proj_dim=None, sample_size=64,
            conditioning_channels=(config.in_channels - latent) if config.is_blobnet else None)

    # ---- weights: packed lazily from the host state dict + the active LoRA adapters
    def effective_state_dict(self):
        """The host state dict with every ACTIVE adapter merged (W + weight * (alpha / r) * B A, weights.merge_lora)."""
        if self._sd is None:
            raise _lib.BlobCtrlHipError("this module was built from packed weights: no host state dict to edit")
        sd = self._sd
        for ad in self._adapters.values():
            if ad["active"] and ad["weight"] != 0.0:
                sd = merge_lora(sd, ad["lora"], ad["alphas"], adapter_scale=ad["weight"])
        return sd

    @property
    def weights(self) -> PackedTrunk:
        if self._packed is None:
            _lib.load()
            self._packed = PackedTrunk(self.effective_state_dict(), self._device, self.trunk_config.block_out_channels)
        return self._packed

    @property
    def packed_fp16(self):
        return self.weights.h_arena

    @property
    def packed_fp32(self):
        return self.weights.f_arena

    def parameters(self, recurse: bool = True):
        """The packed fp16 / fp32 arenas as two frozen Parameters sharing the arenas' storage."""
        w = self.weights
        return iter([torch.nn.Parameter(w.h_arena, requires_grad=False), torch.nn.Parameter(w.f_arena, requires_grad=False)])

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        return iter(zip((prefix + "packed_fp16", prefix + "packed_fp32"), self.parameters()))

    def _invalidate(self):
        """The host weights changed: drop the packed copy and every plan compiled against it (their graphs hold its addresses)."""
        if self._device.type == "cuda" and (self._packed is not None or self._plans):
            torch.cuda.synchronize(self._device)
        for P in self._plans.values():
            P.rec.close()
        self._plans = {}
        self._packed = None
        self._version += 1

    # ---- LoRA (D/loaders/unet.py:271-340 semantics: W + (alpha / r) * B A per target module; merged when the weights are packed)
    def load_lora_adapter(self, lora, alphas, adapter_name="default", weight: float = 1.0):
        if self._sd is None:
            raise _lib.BlobCtrlHipError("this module was built from packed weights: LoRA must be merged before packing")
        missing = [m for m in alphas if m + ".weight" not in self._sd]
        if missing:
            raise KeyError(f"LoRA targets not present in the model: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        if adapter_name in self._adapters:
            raise ValueError(f"Adapter name {adapter_name} already in use in the model - please select a new adapter name.")
        self._adapters[adapter_name] = dict(lora=lora, alphas=alphas, weight=float(weight), active=True)
        self._invalidate()

    def set_adapters(self, adapter_names, weights=None):
        names = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
        ws = [1.0] * len(names) if weights is None else ([weights] * len(names) if not isinstance(weights, (list, tuple)) else list(weights))
        if len(ws) != len(names):
            raise ValueError(f"Length of adapter names {len(names)} is not equal to the length of their weights {len(ws)}.")
        unknown = [n for n in names if n not in self._adapters]
        if unknown:
            raise ValueError(f"Adapter name(s) {set(unknown)} not in the list of present adapters: {set(self._adapters)}.")
        before = [(n, a["active"], a["weight"]) for n, a in self._adapters.items()]
        for n, a in self._adapters.items():
            a["active"] = n in names
        for n, w in zip(names, ws):
            self._adapters[n]["weight"] = 1.0 if w is None else float(w)
        if before != [(n, a["active"], a["weight"]) for n, a in self._adapters.items()]:
            self._invalidate()

    def unload_lora(self):
        if self._adapters:
            self._adapters.clear()
            self._invalidate()

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        """fp32 while the module still holds its host state dict and has not been packed (a freshly loaded checkpoint: the script's
        conv_in surgery `Conv2d(..., dtype=unet.dtype)` then stays in fp32 like the reference's, inf:233-249); fp16 - the compute
        dtype of the packed layouts - once packed."""
        return torch.float32 if (self._sd is not None and self._packed is None) else torch.float16

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
    def __init__(self, state_dict, config: TrunkConfig, device="cuda:0", lazy: bool = False):
        assert config.is_blobnet
        super().__init__(state_dict, config, device, lazy)

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", **_ignored):
        """`BlobNetModel.from_pretrained(blobnet_path, ignore_mismatched_sizes=True)` (inf:252): a directory with
        config.json + diffusion_pytorch_model.safetensors, or the .safetensors file itself."""
        from .checkpoint import load_blobnet
        sd, cfg = load_blobnet(path)
        return cls(sd, cfg, device, lazy=True)

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
        # through the dispatcher (torch.ops.blobctrl.blobnet_forward, ops.py): profilers and fake-tensor tracing see the call
        from . import ops
        outs = torch.ops.blobctrl.blobnet_forward(sample, float(timestep), conditioning_scale, ops.register(self))
        nd = len(ops.blobnet_output_shapes(self.trunk_config, B, H, W)[0])
        return list(outs[:nd]), outs[nd], list(outs[nd + 1:])

    def _forward_impl(self, sample: torch.Tensor, timestep: float, conditioning_scale: float):
        """Body of torch.ops.blobctrl.blobnet_forward: one graph replay of the recorded trunk."""
        B, C, H, W = sample.shape
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
    def __init__(self, state_dict, config: TrunkConfig, device="cuda:0", lazy: bool = False):
        assert not config.is_blobnet
        super().__init__(state_dict, config, device, lazy)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, extra_in_channels=0, lora_path=None, lora_scale=1.0, device="cuda:0", **_ignored):
        """`UNet2DConditionModel.from_pretrained(sd15_path, subfolder="unet")` (inf:229-232): the file's own 4-channel UNet, held on
        the host until first use, so that the script's next statements - the conv_in 4 -> 5 surgery through `unet.conv_in`
        (inf:233-249) and `pipeline.load_lora_weights` (inf:270-273) - work on it unchanged.  `extra_in_channels=1` /
        `lora_path=` do both in this call instead."""
        import os
        from .checkpoint import load_unet
        sd, cfg = load_unet(os.path.join(path, subfolder) if subfolder else path, extra_in_channels, lora_path, lora_scale)
        return cls(sd, cfg, device, lazy=True)

    # ---- `unet.conv_in` as the script uses it (inf:233-249): reads .weight / .bias / .out_channels, assigns a wider Conv2d
    @property
    def conv_in(self):
        if self._sd is None:
            raise _lib.BlobCtrlHipError("this UNet was built from packed weights: conv_in is not editable")
        w, b = self._sd["conv_in.weight"], self._sd.get("conv_in.bias")
        conv = torch.nn.Conv2d(w.shape[1], w.shape[0], kernel_size=3, stride=1, padding=1, bias=b is not None)
        with torch.no_grad():
            conv.weight.copy_(w)
            if b is not None:
                conv.bias.copy_(b)
        return conv

    def __setattr__(self, name, value):
        if name == "conv_in" and isinstance(value, torch.nn.Module):
            if self._sd is None:
                raise _lib.BlobCtrlHipError("this UNet was built from packed weights: conv_in is not editable")
            w = value.weight.detach().float().cpu()
            if w.ndim != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[0] != self._sd["conv_in.weight"].shape[0]:
                raise ValueError(f"conv_in must be a 3x3 convolution with {self._sd['conv_in.weight'].shape[0]} output channels")
            self._sd["conv_in.weight"] = w.clone()
            if value.bias is not None:
                self._sd["conv_in.bias"] = value.bias.detach().float().cpu().clone()
            import dataclasses
            self.trunk_config = dataclasses.replace(self.trunk_config, in_channels=w.shape[1])     # (the caller's config object is not touched)
            self._make_config()
            self._invalidate()
            return
        super().__setattr__(name, value)

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
        # through the dispatcher (torch.ops.blobctrl.unet_forward, ops.py)
        from . import ops
        eps = torch.ops.blobctrl.unet_forward(sample, float(timestep), encoder_hidden_states,
                                              list(down_block_add_samples) if is_blobnet else [], mid_block_add_sample if is_blobnet else None,
                                              list(up_block_add_samples) if is_blobnet else [], ops.register(self))
        if is_blobnet:                                   # the reference consumes the two lists with pop(0) (:1217, 1230, 1313)
            del down_block_add_samples[:]
            del up_block_add_samples[:]
        dt = sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float32
        return (eps.to(dt),)

    def _forward_impl(self, sample, timestep, encoder_hidden_states, down_block_add_samples, mid_block_add_sample, up_block_add_samples):
        """Body of torch.ops.blobctrl.unet_forward: eps [B][4][H][W] fp32."""
        B, C, H, W = sample.shape
        is_blobnet = mid_block_add_sample is not None
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
        out = torch.empty(B, self.trunk_config.out_channels, H, W, dtype=torch.float32, device=self.device)
        out.copy_(P.eps.view(B, H, W, self.trunk_config.out_channels).permute(0, 3, 1, 2))
        return out

