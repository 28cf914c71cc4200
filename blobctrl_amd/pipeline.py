"""The blob-conditioned denoising loop on MI355X: host-side mirror of StableDiffusionBlobNetPipeline.__call__.

Mirrors blobctrl/pipelines/pipeline_blobnet.py:743-1166 for the hot-path part:
same keyword names and meaning (`num_inference_steps`, `guidance_scale`, `generator`, `latents`,
`blobnet_conditioning_scale` (must be a Python float, pipe:395-396), `blobnet_control_guidance_start/end`,
`output_type`), same error behaviour for bad arguments.  The VAE either side of the loop is optional:
with `vae=blobctrl_amd.vae.AutoencoderKL(...)` the call also accepts `fg_image` / `bg_image` (pipe:970-971) and returns decoded
images for `output_type="pt" | "np"` (pipe:1132-1146); with `text_encoder=blobctrl_amd.clip_text.CLIPTextModel(...)`,
`encode_prompt(prompt_ids, negative_prompt_ids)` produces the prompt embeddings (tokenisation stays on the host).

Execution model: one static launch plan per (batch, canvas, steps) configuration -
    prologue (once per edit): cross-attention K/V of the prompt embeddings
    step A (BlobNet active):  assemble BlobNet input -> BlobNet (batch B, CFG halves share it) -> assemble UNet input
                              -> UNet (batch 2B, BlobNet residuals added in GEMM epilogues) -> crop + CFG + scheduler step
    step I (BlobNet inactive, cond_scale * keep[i] == 0): UNet only
each captured once into a hipGraph and replayed per step; per-step scalars (timestep, scheduler coefficients,
conditioning scale) live in device tables indexed by a device-side step counter.
"""
import os
from typing import List, Optional, Union

import torch

from . import _lib
from .engine import TrunkConfig, TrunkPlan
from .launch import Recorder
from .schedulers import DDIMTable, UniPCTable
from .weights import PackedTrunk, pad8


def blobnet_keep(num_steps, start, end):
    """pipe:1006-1012."""
    return [1.0 - float(i / num_steps < start or (i + 1) / num_steps > end) for i in range(num_steps)]


class BlobCtrlEngine:
    """The denoising hot path on MI355X at tensor level (prompt embeddings, image latents, scores, DINO feature in; latents out):
    static launch plans per (batch, canvas, steps), hipGraph replay, two streams.  `StableDiffusionBlobNetPipeline` below wraps it
    with the reference pipeline's constructor and `__call__` signature."""

    def __init__(self, unet_state_dict, blobnet_state_dict, unet_config: TrunkConfig, blobnet_config: TrunkConfig,
                 device="cuda:0", scheduler: str = "unipc", use_graphs: bool = True, vae=None, text_encoder=None,
                 max_cached_plans: int = 4, compile_only: bool = False):
        """`compile_only=True` (device may then be "cpu"): the engine only COMPILES plans into `.bcplan` files (`compile_plan`) for the
        C plan runtime (bc_plan_load / bc_step); nothing can be executed and no GPU is touched."""
        self.device = torch.device(device)
        self.compile_only = compile_only
        if self.device.type != "cuda" and not compile_only:
            raise _lib.BlobCtrlHipError("blobctrl_amd runs on MI355X only (device must be cuda:N); there is no CPU fallback")
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        self.lib = _lib.load()
        self.unet_cfg, self.blob_cfg = unet_config, blobnet_config
        # either reference-schema state dicts (packed here) or already-packed / broadcast replicas (dist.broadcast_packed)
        self.unet_w = unet_state_dict if isinstance(unet_state_dict, PackedTrunk) else \
            PackedTrunk(unet_state_dict, self.device, unet_config.block_out_channels)
        self.blob_w = blobnet_state_dict if isinstance(blobnet_state_dict, PackedTrunk) else \
            PackedTrunk(blobnet_state_dict, self.device, blobnet_config.block_out_channels)
        self.scheduler_kind = scheduler
        self.scheduler_params = (1000, 0.00085, 0.012)               # (num_train_timesteps, beta_start, beta_end): SD-1.5 values
        self.use_graphs = use_graphs and not os.environ.get("BC_NO_GRAPHS")    # diagnostics: eager launches from the host loop
        if self.device.type == "cuda":
            # (stream priorities were measured and do nothing on this part: DESIGN 9)
            self.stream = torch.cuda.Stream(device=self.device)
            self.side_stream = torch.cuda.Stream(device=self.device)     # BlobNet branch runs here, concurrently with the UNet
            self.side_stream2 = torch.cuda.Stream(device=self.device)    # (BC_SPLIT_CFG: the cond half of the UNet batch)
        self.two_streams = not os.environ.get("BC_ONE_STREAM")
        self.loop_graph = os.environ.get("BC_LOOP_GRAPH", "1") != "0"     # whole-edit graph (one launch per edit) vs one graph per step
        self._plans = {}                                              # (batch, canvas, steps, ...) -> plan, least recently used first
        # plan / graph cache traffic (bench.py prints it: C4's eight per-request batches of a rank must replay ONE captured whole-edit graph)
        self.cache_stats = dict(plans_recorded=0, plan_hits=0, loop_graph_captures=0, loop_graph_hits=0)
        self.max_cached_plans = max(1, int(max_cached_plans))         # a 512^2 batch-1 plan holds ~2.5 GB of activations
        self.max_loop_graphs = 4                                      # whole-edit graphs kept per plan (one per active / inactive pattern)
        self._sched_cache = {}
        self.feat_dim = blobnet_config.in_channels - 5
        self.vae = vae                                                # optional blobctrl_amd.vae.AutoencoderKL
        self.text_encoder = text_encoder                              # optional blobctrl_amd.clip_text.CLIPTextModel

    # ------------------------------------------------------------------------------------------------ planning
    def _plan(self, B, h, w, T, ctx_dim, nsteps, per_request=False):
        """per_request: the B samples are B independent edit requests (own fg / bg latents, scores, DINO features and
        conditioning scales) instead of B variations of one edit."""
        key = (B, h, w, T, ctx_dim, nsteps, per_request)
        if key in self._plans:
            self._plans[key] = self._plans.pop(key)                   # mark as most recently used
            self.cache_stats["plan_hits"] += 1
            return self._plans[key]
        self.cache_stats["plans_recorded"] += 1
        while len(self._plans) >= self.max_cached_plans:              # evict the least recently used plan and its graphs
            old = self._plans.pop(next(iter(self._plans)))
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            old.rec.close()                                           # graphs and events of the evicted plan
        dev = self.device
        rec = Recorder(dev)
        P = type("Plan", (), {})()
        P.rec = rec
        H, W = h, 2 * w
        F = self.feat_dim
        f32 = torch.float32
        Bi = B if per_request else 1                         # images (fg / bg / score / feature sets) behind the batch
        P.Bi = Bi
        P.latents = rec.zeros(B, 4, h, w, dtype=f32, name="latents")
        P.fg_lat = rec.zeros(Bi, 4, h, w, dtype=f32, name="fg_lat")
        P.bg_lat = rec.zeros(Bi, 4, h, w, dtype=f32, name="bg_lat")
        P.fg_score = rec.zeros(Bi, h, w, dtype=f32, name="fg_score")
        P.bg_score = rec.zeros(Bi, h, w, dtype=f32, name="bg_score")
        P.feat = rec.zeros(Bi, max(F, 1), dtype=f32, name="feat")
        P.ctx = rec.zeros(2 * B, T, ctx_dim, name="ctx")
        P.step_idx = rec.zeros(1, dtype=torch.int32, name="step_idx")
        P.t_table = rec.zeros(nsteps, dtype=f32, name="t_table")
        P.coef = rec.zeros(nsteps, 16, dtype=f32, name="coef")
        P.scale_table = rec.zeros(nsteps * Bi, dtype=f32, name="scale_table")   # [step][image] conditioning_scale * keep
        P.hist = rec.zeros(3, B * 4 * h * w, dtype=f32, name="hist")
        P.eps_guided = rec.zeros(B, 4, h, w, dtype=f32, name="eps_guided")
        P.guidance = [7.5]

        # rank-1 collapse of the BlobNet feature channels (per-edit weight; round 6: per-IMAGE weights for a batch of independent requests -
        # conv_in then runs as one launch per image, engine.record_collapse)
        no_im2col = False
        P.collapse = F > 3 and "conv_in.featmat" in self.blob_w.h and not no_im2col
        unet_cin = pad8(self.unet_cfg.in_channels)
        blob_cin = 8 if P.collapse else pad8(self.blob_cfg.in_channels)
        # 8-channel inputs (the UNet's 4 latents + score; BlobNet's rank-1-collapsed 4 latents + 2 x score) are assembled directly as
        # the 3x3 im2col operand [rows][128] of conv_in, which then runs as a dense GEMM on the LDS-DMA fast path (K = 72 does not)
        P.unet_im2col = unet_cin == 8 and not no_im2col
        P.blob_im2col = P.collapse
        P.feat16 = rec.zeros(Bi, pad8(max(F, 1)), name="feat16")
        P.blob_in = rec.zeros(B, H * W, 128 if P.blob_im2col else blob_cin)
        P.unet_in = rec.zeros(2 * B, H * W, 128 if P.unet_im2col else unet_cin)

        # ---- prologue: prompt K/V
        P.prologue = rec.begin("prologue")
        unet_a = TrunkPlan(rec, self.unet_w, self.unet_cfg, 2 * B, H, W)
        unet_a.record_context(P.ctx, T)
        blob = TrunkPlan(rec, self.blob_w, self.blob_cfg, B, H, W)
        if P.collapse:
            blob.record_collapse(P.feat16, per_image=B if per_request else 0)
        # time-embedding path of every step, once per edit (read in the step through the device step counter)
        temb_per_step = False                                # (the per-edit table replaced the four launches per net inside every step)
        if not temb_per_step:
            unet_a.record_time_table(P.t_table, nsteps, P.step_idx)
            blob.record_time_table(P.t_table, nsteps, P.step_idx)

        # CFG halves on two streams (opt-in, BC_SPLIT_CFG=1; single edits only - larger batches fill the GPU with the batch-2B UNet)
        split_cfg = B == 1 and self.two_streams and not temb_per_step and bool(os.environ.get("BC_SPLIT_CFG"))
        P.split_cfg = split_cfg
        if split_cfg:
            # the uncond / cond halves of the UNet batch as two batch-B plans on their own streams (ids 0 and 2): more concurrent
            # kernels for a GPU that one batch-2 UNet does not fill, at the price of reading the UNet weights twice per step
            # (measured, same box, interleaved: 11.56 vs 11.73 ms per active step, +2-3 % edits/s; 1108 instead of 725 launches.
            #  Not the default: it trades per-kernel efficiency - every self-attention then runs at batch 1, the 64x64-tile GEMM
            #  becomes the largest kernel of the step - for concurrency, a gain inside the box-to-box spread)
            half_time = TrunkPlan(rec, self.unet_w, self.unet_cfg, B, H, W)
            half_time.record_time_table(P.t_table, nsteps, P.step_idx)
            P.eps_all = rec.zeros(2 * B, H * W, self.unet_cfg.out_channels, dtype=f32)

        def half_plan(b0):
            hp = TrunkPlan(rec, self.unet_w, self.unet_cfg, B, H, W)
            hp.ctx_kv = {bp: (ck[b0 * T:(b0 + B) * T], cvt[b0:b0 + B], T_, ld) for bp, (ck, cvt, T_, ld) in unet_a.ctx_kv.items()}
            # (the K / V^T fragment streams of CHAIN_MIDX are laid out per image too: without the slice both halves fell back to the
            #  unfused MID + attention launches)
            hp.ctx_kvs = {bp: kvs.view(2 * B, -1)[b0:b0 + B].reshape(-1) for bp, kvs in getattr(unet_a, "ctx_kvs", {}).items()}
            hp.ctx_folded = {bp: tuple(t[b0:b0 + B] for t in f) for bp, f in getattr(unet_a, "ctx_folded", {}).items()}
            hp.tproj, hp.tproj_table = half_time.tproj, half_time.tproj_table
            return hp

        def record_unet(plan, residuals):
            if P.unet_im2col:
                rec.call("bc_assemble_input_im2col", P.latents.data_ptr(), B, P.bg_lat.data_ptr(), P.bg_score.data_ptr(), Bi, 2 * B, h, w,
                         0, P.unet_in.data_ptr(), kind="assemble")
            else:
                rec.call("bc_assemble_input", P.latents.data_ptr(), B, P.bg_lat.data_ptr(), P.bg_score.data_ptr(), None, Bi, 0,
                         2 * B, h, w, unet_cin, 0, P.unet_in.data_ptr(), kind="assemble")
            if temb_per_step:
                plan.record_time(P.t_table, P.step_idx)
            if split_cfg:
                ready, joined = rec.new_event(), rec.new_event()
                rec.signal(ready)                                   # unet_in assembled (stream 0)
                rec.sid = 2
                rec.wait(ready)
                half_plan(B).record_forward(P.unet_in[B:], residuals, eps_out=P.eps_all[B:], im2col=P.unet_im2col)   # cond half
                rec.signal(joined)
                rec.sid = 0
                half_plan(0).record_forward(P.unet_in[:B], residuals, eps_out=P.eps_all[:B], im2col=P.unet_im2col)   # uncond half
                rec.wait(joined)
                eps = P.eps_all
            else:
                eps = plan.record_forward(P.unet_in, residuals, im2col=P.unet_im2col, cfg_pairs=True)   # (images b and b + B: the CFG pair)
            P.eps = eps
            rec.call("bc_cfg_scheduler_step", eps, P.latents, P.coef, P.step_idx, P.hist, -1.0, B, h, w, P.eps_guided, 1,
                     kind="cfg_step")

        # ---- step A: BlobNet + UNet
        # The BlobNet branch is recorded for the side stream: fork (side waits for the start of the step on main), every
        # residual is signalled when its zero-conv finishes, the UNet waits right before the GEMM whose epilogue adds it.
        P.step_active = rec.begin("step_active")
        fork = rec.new_event()
        rec.signal(fork)
        rec.sid = 1
        rec.wait(fork)
        if P.blob_im2col:
            rec.call("bc_assemble_input_im2col", P.latents.data_ptr(), B, P.fg_lat.data_ptr(), P.fg_score.data_ptr(), Bi, B, h, w, 1,
                     P.blob_in.data_ptr(), kind="assemble")
        else:
            rec.call("bc_assemble_input", P.latents.data_ptr(), B, P.fg_lat.data_ptr(), P.fg_score.data_ptr(),
                     P.feat.data_ptr() if F > 0 else None, Bi, F, B, h, w, blob_cin, 0, P.blob_in.data_ptr(), kind="assemble")
        if temb_per_step:
            blob.record_time(P.t_table, P.step_idx)
        residuals = blob.record_forward(P.blob_in, None, zero_scale=(1.0, P.scale_table, P.step_idx, B if per_request else 0),
                                        signal_residuals=True, im2col=P.blob_im2col)
        rec.sid = 0
        P.residuals = residuals
        record_unet(unet_a, residuals)
        P.eps_active = P.eps

        # ---- step I: UNet only (cond_scale == 0: BlobNet output is multiplied by 0, bn:936-938)
        P.step_inactive = rec.begin("step_inactive")
        unet_i = TrunkPlan(rec, self.unet_w, self.unet_cfg, 2 * B, H, W)
        unet_i.ctx_kv, unet_i.ctx_kvs, unet_i.ctx_folded = unet_a.ctx_kv, getattr(unet_a, "ctx_kvs", {}), getattr(unet_a, "ctx_folded", {})
        if not temb_per_step:
            unet_i.tproj, unet_i.tproj_table = unet_a.tproj, unet_a.tproj_table
        record_unet(unet_i, None)
        P.eps_inactive = P.eps
        P.loop_graphs = {}
        P.captured = False
        from . import options
        P.options, P.options_non_default = options.effective(), options.non_default()     # (what BC_PLAN said while this plan was recorded)
        self._plans[key] = P
        return P

    def set_scheduler(self, kind, params=None):
        """`kind` "unipc" | "ddim"; `params` = (num_train_timesteps, beta_start, beta_end) of the scheduler's configuration (the
        drop-in scheduler objects accept non-default betas: the engine must tabulate the SAME alphas)."""
        if kind not in ("unipc", "ddim"):
            raise NotImplementedError(f"scheduler {kind!r} has no coefficient table (UniPC and DDIM have)")
        self.scheduler_kind = kind
        if params is not None:
            self.scheduler_params = (int(params[0]), float(params[1]), float(params[2]))

    def _scheduler_table(self, n):
        """Coefficient tables depend only on (scheduler, its beta configuration, steps)."""
        key = (self.scheduler_kind, self.scheduler_params, n)
        sched = self._sched_cache.get(key)
        if sched is None:
            nt, b0, b1 = self.scheduler_params
            cls = UniPCTable if self.scheduler_kind == "unipc" else DDIMTable
            sched = cls(num_train_timesteps=nt, beta_start=b0, beta_end=b1)
            sched.set_timesteps(n)
            self._sched_cache[key] = sched
        return sched

    def set_weights(self, unet_w=None, blob_w=None):
        """New packed weights (LoRA loaded / unloaded, conv_in edited): every cached plan and its graphs hold the old addresses."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for P in self._plans.values():
            P.rec.close()
        self._plans = {}
        if unet_w is not None:
            self.unet_w = unet_w
        if blob_w is not None:
            self.blob_w = blob_w
            self.feat_dim = self.blob_cfg.in_channels - 5

    def _capture(self, P):
        if P.captured or not self.use_graphs:
            return
        s, side = self._streams()
        # warm-up run outside capture (module loading, attribute setting) then capture each segment once
        torch.cuda.synchronize(self.device)
        P.prologue.run(s)
        for seg in (P.step_active, P.step_inactive):
            with torch.cuda.stream(self.stream):
                P.step_idx.zero_()
            seg.run(s, side, self._extra())
        torch.cuda.synchronize(self.device)
        for seg in (P.step_active, P.step_inactive):
            seg.capture(s, side, self._extra())
        torch.cuda.synchronize(self.device)
        P.captured = True

    def _join_caller(self):
        """The engine works on private non-blocking streams: make them wait for whatever the caller has enqueued on ITS current
        stream (inputs produced by splat_features / Dinov2Model / torch ops are asynchronous) before reading any input."""
        cur = torch.cuda.current_stream(self.device)
        for st in (self.stream, self.side_stream, self.side_stream2):
            st.wait_stream(cur)

    def _streams(self):
        s = self.stream.cuda_stream
        return s, (self.side_stream.cuda_stream if self.two_streams else s)

    def _extra(self):
        return (self.side_stream2.cuda_stream if self.two_streams else self.stream.cuda_stream,)

    # ------------------------------------------------------------------------------------------------ call
    def check_inputs(self, blobnet_conditioning_scale, start, end, num_inference_steps):
        if isinstance(blobnet_conditioning_scale, (list, tuple)):                   # per-request batch (extension)
            if not all(isinstance(v, float) for v in blobnet_conditioning_scale):
                raise TypeError("per-request `blobnet_conditioning_scale` must be a list of `float`.")
        elif not isinstance(blobnet_conditioning_scale, float):                     # pipe:395-396
            raise TypeError("For single blobnet: `blobnet_conditioning_scale` must be type `float`.")
        if start >= end:                                                            # pipe:424-427
            raise ValueError(f"control guidance start: {start} cannot be larger or equal to control guidance end: {end}.")
        if start < 0.0:
            raise ValueError(f"control guidance start: {start} can't be smaller than 0.")
        if end > 1.0:
            raise ValueError(f"control guidance end: {end} can't be larger than 1.0.")
        if num_inference_steps < 1:
            raise ValueError("num_inference_steps must be >= 1")

    def encode_prompt(self, prompt_ids: torch.Tensor, negative_prompt_ids: torch.Tensor, clip_skip: Optional[int] = None):
        """pipe:508-687 after tokenisation: token ids [B, 77] of the prompt and of the negative prompt -> prompt_embeds
        [2B, 77, D] = cat(negative, positive) (pipe:937-949), ready for `__call__`."""
        if self.text_encoder is None:
            raise ValueError("this pipeline was built without a text encoder: pass prompt_embeds, or text_encoder=...")
        if prompt_ids.shape != negative_prompt_ids.shape:
            raise ValueError("prompt_ids and negative_prompt_ids must have the same shape")
        self._join_caller()
        with torch.cuda.stream(self.stream):
            emb = self.text_encoder(torch.cat([negative_prompt_ids, prompt_ids], 0), clip_skip=clip_skip)[0]
        self.stream.synchronize()
        return emb

    def encode_latents(self, image: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """pipe:300-309: image [1,3,H,W] in [-1,1] -> posterior sample * scaling_factor, [1,4,H/8,W/8] fp32."""
        if self.vae is None:
            raise ValueError("this pipeline was built without a VAE: pass fg_image_latents / bg_image_latents, or vae=...")
        self._join_caller()
        with torch.cuda.stream(self.stream):
            lat = self.vae.encode(image).latent_dist.sample(generator, scale=self.vae.config.scaling_factor)
        self.stream.synchronize()
        return lat

    def decode_latents(self, latents: torch.Tensor, output_type: str = "pt"):
        """pipe:1132-1146: vae.decode(latents / scaling_factor) then VaeImageProcessor.postprocess with denormalisation
        ((x/2+0.5).clamp(0,1); "pt" = [B,3,H,W] tensor, "np" = [B,H,W,3] numpy)."""
        if self.vae is None:
            raise ValueError("this pipeline was built without a VAE: use output_type='latent', or pass vae=...")
        self._join_caller()
        with torch.cuda.stream(self.stream):
            img = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]
            img = (img / 2 + 0.5).clamp(0, 1)
        self.stream.synchronize()
        return img if output_type == "pt" else img.permute(0, 2, 3, 1).cpu().float().numpy()

    @torch.no_grad()
    def denoise(self, prompt_embeds: torch.Tensor, fg_image_latents: Optional[torch.Tensor] = None,
                 bg_image_latents: Optional[torch.Tensor] = None, gs_score: torch.Tensor = None, dino_feats: Optional[torch.Tensor] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, generator: Optional[torch.Generator] = None,
                 latents: Optional[torch.Tensor] = None, blobnet_conditioning_scale: float = 1.0,
                 blobnet_control_guidance_start: float = 0.0, blobnet_control_guidance_end: float = 1.0,
                 output_type: str = "latent", callback_on_step_end=None, trace: Optional[list] = None,
                 teacher_latents: Optional[List[torch.Tensor]] = None, fg_image: Optional[torch.Tensor] = None,
                 bg_image: Optional[torch.Tensor] = None, return_sample: bool = False, eta: float = 0.0,
                 do_classifier_free_guidance: Optional[bool] = None, callback_self=None):
        """prompt_embeds [2B, T, D] = cat(negative, positive) (pipe:937-949); fg/bg_image_latents [1,4,h,w] already scaled
        by 0.18215 (pipe:300-309); gs_score [1,2,h,w] = (bg, fg) scores (pipe:974); dino_feats [1,1,F] (pipe:982).
        Instead of the latents, `fg_image` / `bg_image` [1,3,8h,8w] in [-1,1] may be given when the pipeline has a VAE.
        Returns the final latents [B,4,h,w] fp32 for `output_type="latent"` (pipe:1143), else the decoded, denormalised
        images ("pt" / "np", pipe:1132-1146)."""
        if return_sample:
            # pipe:1052-1061 reads blobnet.conv_norm_out / conv_out, which BlobNetModel does not have (626-tensor schema): dead code
            raise NotImplementedError("return_sample=True is not supported (the reference path dereferences layers BlobNet lacks)")
        if eta != 0.0:
            raise NotImplementedError("eta != 0 (stochastic DDIM) is not supported: the scheduler tables are the eta = 0 update")
        if output_type not in ("latent", "pt", "np"):
            raise ValueError(f"output_type must be 'latent', 'pt' or 'np', got {output_type!r}")
        if output_type != "latent" and self.vae is None:
            raise ValueError("this pipeline was built without a VAE: use output_type='latent', or pass vae=...")
        if gs_score is None:
            raise ValueError("gs_score is required")
        if fg_image_latents is None:
            if fg_image is None:
                raise ValueError("give fg_image_latents or fg_image")
            fg_image_latents = self.encode_latents(fg_image)
        if bg_image_latents is None:
            if bg_image is None:
                raise ValueError("give bg_image_latents or bg_image")
            bg_image_latents = self.encode_latents(bg_image)
        self.check_inputs(blobnet_conditioning_scale, blobnet_control_guidance_start, blobnet_control_guidance_end,
                          num_inference_steps)
        # pipe:494-497: guidance_scale <= 1 switches classifier-free guidance OFF in the reference (prompt_embeds then holds the
        # positive prompt only and the UNet output is used as is).  The engine keeps its CFG-batch-2 plan: the positive embeddings
        # fill both halves and the effective scale is 1, eps_u + 1 * (eps_c - eps_u) = eps_c.
        # `do_classifier_free_guidance=False` (what the reference derives from guidance_scale <= 1) says explicitly that prompt_embeds
        # holds the positive prompts only; without the flag the layout is inferred from `latents` and refused when ambiguous.
        if do_classifier_free_guidance is None:
            do_classifier_free_guidance = guidance_scale > 1.0
            if not do_classifier_free_guidance:
                if latents is None:
                    raise ValueError("guidance_scale <= 1 without `latents`: pass do_classifier_free_guidance=False (prompt_embeds = "
                                     "positive prompts only) or =True (negative and positive halves) - the layout cannot be inferred")
                do_classifier_free_guidance = prompt_embeds.shape[0] != latents.shape[0]
        if not do_classifier_free_guidance:
            prompt_embeds = torch.cat([prompt_embeds, prompt_embeds], 0)
        if guidance_scale <= 1.0 or not do_classifier_free_guidance:
            guidance_scale = 1.0
        B2, T, Dc = prompt_embeds.shape
        if B2 % 2:
            raise ValueError("prompt_embeds must hold the negative and positive halves (classifier-free guidance)")
        B = B2 // 2
        h, w = fg_image_latents.shape[-2:]
        n = num_inference_steps
        per_request = fg_image_latents.dim() == 4 and fg_image_latents.shape[0] > 1
        Bi = B if per_request else 1
        if per_request:
            for name, t_ in (("fg_image_latents", fg_image_latents), ("bg_image_latents", bg_image_latents), ("gs_score", gs_score),
                             ("dino_feats", dino_feats)):
                if t_ is not None and t_.shape[0] != B:
                    raise ValueError(f"request batch: {name} must have leading dimension {B} (one per request), got {t_.shape[0]}")
        req_scales = list(blobnet_conditioning_scale) if isinstance(blobnet_conditioning_scale, (list, tuple)) else \
            [blobnet_conditioning_scale] * Bi
        if len(req_scales) != Bi:
            raise ValueError(f"blobnet_conditioning_scale: expected {Bi} values, got {len(req_scales)}")
        P = self._plan(B, h, w, T, Dc, n, per_request)
        dev = self.device
        sched = self._scheduler_table(n)
        self.timesteps = sched.timesteps
        if latents is None:                                                          # pipe:438-453
            latents = torch.randn((B, 4, h, w), generator=generator, device=generator.device if generator else "cpu",
                                  dtype=torch.float32)
        keep = blobnet_keep(n, blobnet_control_guidance_start, blobnet_control_guidance_end)
        scale_rows = [[sc * k for sc in req_scales] for k in keep]                   # [step][image]
        scales = [max(abs(v) for v in row) for row in scale_rows]                    # a step is BlobNet-free iff every scale is 0
        bg, fg = gs_score.unbind(dim=1)                                              # pipe:974
        self._join_caller()
        with torch.cuda.stream(self.stream):
            P.latents.copy_(latents.to(dev, torch.float32) * sched.init_noise_sigma)
            P.fg_lat.copy_(fg_image_latents.to(dev, torch.float32).reshape(Bi, 4, h, w))
            P.bg_lat.copy_(bg_image_latents.to(dev, torch.float32).reshape(Bi, 4, h, w))
            P.fg_score.copy_(fg.to(dev, torch.float32).reshape(Bi, h, w))
            P.bg_score.copy_(bg.to(dev, torch.float32).reshape(Bi, h, w))
            if self.feat_dim > 0:
                if dino_feats is None:
                    raise ValueError("dino_feats is required (BlobNet conditioning channels)")
                P.feat.copy_(dino_feats.to(dev, torch.float32).reshape(Bi, self.feat_dim))
                if P.collapse:
                    P.feat16[:, : self.feat_dim].copy_(P.feat)
            P.ctx.copy_(prompt_embeds.to(dev, torch.float16))
            P.t_table.copy_(sched.timesteps.to(torch.float32))
            coef = sched.table().clone()
            coef[:, 11] = float(guidance_scale)          # read by the captured cfg/scheduler kernel
            P.coef.copy_(coef)
            P.scale_table.copy_(torch.tensor(scale_rows, dtype=torch.float32).reshape(-1))
            P.step_idx.zero_()
            P.hist.zero_()
        P.guidance[0] = float(guidance_scale)
        s, side = self._streams()
        if teacher_latents is None and callback_on_step_end is None and trace is None and not P.captured:
            self.stream.synchronize()
            self._capture(P)
            # captured segments left the step counter advanced by the warm-up runs: reset per-edit state
            with torch.cuda.stream(self.stream):
                P.latents.copy_(latents.to(dev, torch.float32) * sched.init_noise_sigma)
                P.step_idx.zero_()
                P.hist.zero_()
        plain = teacher_latents is None and callback_on_step_end is None and trace is None
        if plain and self.use_graphs and self.loop_graph:
            # the WHOLE edit (prologue + n steps) as ONE hipGraph per (plan, active / inactive pattern): one launch per edit; the
            # per-step scalars (timestep, scheduler coefficients, conditioning scale) are read through the device step counter
            key = tuple(v != 0.0 for v in scales)
            g = P.loop_graphs.pop(key, None)
            self.cache_stats["loop_graph_hits" if g is not None else "loop_graph_captures"] += 1
            if g is None:
                torch.cuda.synchronize(self.device)
                while len(P.loop_graphs) >= self.max_loop_graphs:              # least recently used pattern: destroy its graph exec
                    P.rec.destroy_loop_graph(P.loop_graphs.pop(next(iter(P.loop_graphs))))
                segs = [P.prologue] + [P.step_active if a else P.step_inactive for a in key]
                g = P.rec.capture_loop(segs, s, side, self._extra())
            P.loop_graphs[key] = g                                             # (re-inserted: most recently used last)
            _lib.check(self.lib.bc_graph_launch(g, s), "bc_graph_launch")
            return self._result(P, output_type)
        P.prologue.run(s)
        for i in range(n):
            if teacher_latents is not None:
                with torch.cuda.stream(self.stream):
                    P.latents.copy_(teacher_latents[i].to(dev, torch.float32))
            seg = P.step_active if scales[i] != 0.0 else P.step_inactive
            seg.run(s, side, self._extra())
            if trace is not None:
                self.stream.synchronize()
                trace.append((P.eps_guided.clone(), P.latents.clone()))
            if callback_on_step_end is not None:
                self.stream.synchronize()
                ret = callback_on_step_end(callback_self or self, i, int(sched.timesteps[i]), {"latents": P.latents})
                if isinstance(ret, dict) and ret.get("latents") is not None and ret["latents"] is not P.latents:
                    with torch.cuda.stream(self.stream):                         # pipe:1112 `latents = callback_outputs.pop(...)`
                        P.latents.copy_(ret["latents"].to(dev, torch.float32))
        return self._result(P, output_type)

    def _result(self, P, output_type):
        """The final latents as a fresh tensor with ordinary stream semantics: the copy is enqueued behind the edit on the engine's
        stream and the CALLER's current stream is made to wait for it - no host synchronisation here, so the next edit's inputs and
        graph launch are enqueued while this edit's tail still runs (a host sync per edit drained the device for ~20 ms of a 550 ms
        edit).  Reading the tensor (`.cpu()`, `torch.cuda.synchronize()`) synchronises as for any torch op."""
        with torch.cuda.stream(self.stream):
            out = P.latents.clone()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        out.record_stream(cur)                              # (allocated on the engine's stream, consumed on the caller's)
        return out if output_type == "latent" else self.decode_latents(out, output_type)

    @torch.no_grad()
    def __call__(self, prompt_embeds, fg_image_latents=None, bg_image_latents=None, gs_score=None, dino_feats=None, **kw):
        """`denoise(...)`; a plain tensor-level edit (explicit start latents, latent output, no callbacks / traces / images) goes through
        the dispatcher as torch.ops.blobctrl.denoise (ops.py), everything else calls `denoise` directly."""
        plain = {"num_inference_steps", "guidance_scale", "latents", "blobnet_conditioning_scale", "blobnet_control_guidance_start",
                 "blobnet_control_guidance_end"}
        sc = kw.get("blobnet_conditioning_scale", 1.0)
        # (ints and anything else take `denoise` directly, which raises the reference's TypeError for them)
        is_list = isinstance(sc, (list, tuple))
        sc_ok = isinstance(sc, float) or (is_list and len(sc) > 0 and all(isinstance(v, float) for v in sc))
        if (set(kw) <= plain and kw.get("latents") is not None and fg_image_latents is not None and bg_image_latents is not None
                and gs_score is not None and dino_feats is not None and kw.get("guidance_scale", 7.5) > 1.0 and sc_ok):   # (else: denoise raises)
            from . import ops
            return torch.ops.blobctrl.denoise(prompt_embeds, fg_image_latents, bg_image_latents, gs_score, dino_feats, kw["latents"],
                                              int(kw.get("num_inference_steps", 50)), float(kw.get("guidance_scale", 7.5)),
                                              [float(v) for v in sc] if is_list else [float(sc)],
                                              float(kw.get("blobnet_control_guidance_start", 0.0)),
                                              float(kw.get("blobnet_control_guidance_end", 1.0)), ops.register(self), is_list)
        return self.denoise(prompt_embeds, fg_image_latents, bg_image_latents, gs_score, dino_feats, **kw)

    def compile_plan(self, path, B, h, w, T, ctx_dim, num_inference_steps, guidance_scale=7.5, blobnet_conditioning_scale=1.0,
                     blobnet_control_guidance_start=0.0, blobnet_control_guidance_end=1.0):
        """Write the launch plan of one edit configuration as a relocatable `.bcplan` file for the C plan runtime
        (include/blobctrl_hip.h: bc_plan_load / bc_plan_buffer / bc_step / bc_plan_capture_loop): segments "prologue",
        "step_active", "step_inactive"; packed weights and the scheduler / guidance tables stored with their contents; the per-edit
        inputs are the NAMED buffers latents, fg_lat, bg_lat, fg_score, bg_score (fp32), feat (fp32) / feat16 (fp16, rank-1
        collapse), ctx (fp16 [2B][T][ctx_dim] = cat(negative, positive)); the result is read from `latents`.  Returns the per-step
        segment names (which steps run BlobNet)."""
        n = num_inference_steps
        P = self._plan(B, h, w, T, ctx_dim, n, False)
        sched = self._scheduler_table(n)
        keep = blobnet_keep(n, blobnet_control_guidance_start, blobnet_control_guidance_end)
        P.t_table.copy_(sched.timesteps.to(torch.float32))
        coef = sched.table().clone()
        coef[:, 11] = float(guidance_scale)
        P.coef.copy_(coef)
        P.scale_table.copy_(torch.tensor([blobnet_conditioning_scale * k for k in keep], dtype=torch.float32))
        for t in (P.t_table, P.coef, P.scale_table):                 # constants of this configuration: saved WITH their contents
            P.rec._workspace.discard(t.untyped_storage().data_ptr())
        P.rec.save(path)
        return ["step_active" if blobnet_conditioning_scale * k != 0.0 else "step_inactive" for k in keep]

    # convenience for bench / tests ------------------------------------------------------------------
    def plan_for(self, B, h, w, T, ctx_dim, nsteps, per_request=False):
        return self._plan(B, h, w, T, ctx_dim, nsteps, per_request)


# ======================================================================================================================
# The reference's pipeline surface: same constructor keywords, same __call__ signature, same error behaviour,
# `.images` output - so that scripts/blobctrl_inference.py:191-205 runs unchanged on the MI355X modules.
# ======================================================================================================================
class StableDiffusionBlobNetPipelineOutput:
    """pipeline_blobnet.py:1163-1166 (`images`, `nsfw_content_detected`; tuple-like for `return_dict=False` callers)."""

    def __init__(self, images, nsfw_content_detected=None):
        self.images, self.nsfw_content_detected = images, nsfw_content_detected

    def __iter__(self):
        return iter((self.images, self.nsfw_content_detected))

    def __getitem__(self, i):
        return (self.images, self.nsfw_content_detected)[i]


class StableDiffusionBlobNetPipeline:
    """Drop-in for blobctrl/pipelines/pipeline_blobnet.py:StableDiffusionBlobNetPipeline on MI355X.

    Components (pipe:206-243): `vae` = blobctrl_amd.vae.AutoencoderKL, `unet` / `blobnet` = blobctrl_amd.modules shells (any LoRA is
    merged when the UNet is packed), `tokenizer` = any callable with the CLIPTokenizer call contract (host side), `text_encoder` =
    blobctrl_amd.clip_text.CLIPTextModel, `scheduler` = blobctrl_amd.schedulers.UniPCMultistepScheduler | DDIMScheduler,
    `dinov2_processor` = image_processor.Dinov2ImageProcessor (or the HF processor), `dinov2` = blobctrl_amd.dinov2.Dinov2Model.
    The denoise loop itself is the captured-plan engine (BlobCtrlEngine): the module shells are not called per step."""

    _callback_tensor_inputs = ["latents", "image_embeds", "negative_image_embeds"]

    def __init__(self, vae, unet, tokenizer, text_encoder, blobnet, scheduler, safety_checker=None, dinov2_processor=None,
                 dinov2=None, requires_safety_checker: bool = False, use_graphs: bool = True):
        from .image_processor import Dinov2ImageProcessor, VaeImageProcessor
        from .schedulers import TableScheduler
        if safety_checker is not None:
            raise NotImplementedError("the reference disables the safety checker (pipe:1133-1135); pass safety_checker=None")
        if not isinstance(scheduler, TableScheduler):
            raise TypeError("scheduler must be blobctrl_amd.schedulers.UniPCMultistepScheduler or DDIMScheduler")
        self.vae, self.unet, self.blobnet, self.tokenizer, self.text_encoder = vae, unet, blobnet, tokenizer, text_encoder
        self.dinov2, self.dinov2_processor = dinov2, dinov2_processor if dinov2_processor is not None else Dinov2ImageProcessor()
        self.safety_checker = None
        self.device = unet.device
        self._use_graphs = use_graphs
        self._engine, self._engine_versions = None, None
        self._scheduler = scheduler
        nblocks = len(getattr(vae.config, "block_out_channels", (0, 0, 0, 0))) if vae is not None else 4
        self.vae_scale_factor = 2 ** (nblocks - 1)                                                  # pipe:241
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True)    # pipe:242
        self._guidance_scale, self._clip_skip, self._num_timesteps = 7.5, None, 0

    # ---- construction as the scripts do it (inf:260-279)
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, unet=None, blobnet=None, dinov2_processor=None, dinov2=None,
                        vae=None, text_encoder=None, tokenizer=None, scheduler=None, safety_checker=None, torch_dtype=None,
                        device="cuda:0", **_ignored):
        """`StableDiffusionBlobNetPipeline.from_pretrained(sd15_path, unet=, blobnet=, torch_dtype=, dinov2_processor=, dinov2=)`
        (inf:260-267; D/pipelines/pipeline_utils.py from_pretrained): components that are not passed are built from the sub-folders
        of the model directory - `vae/`, `text_encoder/`, `tokenizer/` (vocab.json + merges.txt), `scheduler/scheduler_config.json`,
        `unet/` - with this package's classes.  `torch_dtype` is accepted and ignored (the kernels compute in fp16 / fp32
        accumulate); the safety checker stays off as in the reference (pipe:1133-1135)."""
        import os
        from .clip_text import CLIPTextModel
        from .modules import UNet2DConditionModel
        from .schedulers import scheduler_from_config_dir
        from .vae import AutoencoderKL
        root = pretrained_model_name_or_path
        if not os.path.isdir(root):
            raise FileNotFoundError(f"{root}: not a local model directory (there is no hub download here)")
        sub = lambda name: os.path.isdir(os.path.join(root, name))
        dev = str(unet.device) if unet is not None else device
        if unet is None:
            unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet", device=dev)
        if blobnet is None:
            raise ValueError("blobnet= is required (the SD-1.5 directory does not hold one; inf:252)")
        if vae is None and sub("vae"):
            vae = AutoencoderKL.from_pretrained(root, subfolder="vae", device=dev)
        if text_encoder is None and sub("text_encoder"):
            text_encoder = CLIPTextModel.from_pretrained(root, subfolder="text_encoder", device=dev)
        if tokenizer is None and sub("tokenizer"):
            from .clip_tokenizer import CLIPTokenizer
            tokenizer = CLIPTokenizer.from_pretrained(root, subfolder="tokenizer")
        if scheduler is None:
            scheduler = scheduler_from_config_dir(os.path.join(root, "scheduler"))
        return cls(vae=vae, unet=unet, tokenizer=tokenizer, text_encoder=text_encoder, blobnet=blobnet, scheduler=scheduler,
                   safety_checker=None, dinov2_processor=dinov2_processor, dinov2=dinov2)

    @property
    def engine(self) -> BlobCtrlEngine:
        """The captured-plan loop engine over the CURRENT packed weights of `unet` / `blobnet`: built on first use, re-pointed (plans
        dropped) whenever a module's host weights changed (LoRA loaded / unloaded, conv_in surgery)."""
        versions = (self.unet._version, self.blobnet._version)
        if self._scheduler.kind is None:
            # the PNDM configuration holder a checkpoint directory ships (inf:276 replaces it): no loop engine behind it
            raise NotImplementedError(f"{type(self._scheduler).__name__} is a configuration holder here: assign a UniPCMultistepScheduler / "
                                      "DDIMScheduler (e.g. UniPCMultistepScheduler.from_config(pipe.scheduler.config)) before running the loop")
        if self._engine is None:
            kind = self._scheduler.kind
            self._engine = BlobCtrlEngine(self.unet.weights, self.blobnet.weights, self.unet.trunk_config, self.blobnet.trunk_config,
                                          device=str(self.unet.device), scheduler=kind, use_graphs=self._use_graphs, vae=self.vae,
                                          text_encoder=self.text_encoder)
            self._engine_versions = versions
        elif versions != self._engine_versions:
            self._engine.unet_cfg, self._engine.blob_cfg = self.unet.trunk_config, self.blobnet.trunk_config
            self._engine.set_weights(None, None)                     # frees the plans that point into the old arenas first
            self._engine.unet_w = self._engine.blob_w = None
            self._engine.set_weights(self.unet.weights, self.blobnet.weights)
            self._engine_versions = versions
        if self._scheduler.kind is not None:
            self._engine.set_scheduler(self._scheduler.kind, self._scheduler.table_params()
                                       if hasattr(self._scheduler, "table_params") else None)
        return self._engine

    # ---- attribute plumbing the scripts touch (inf:276-279)
    @property
    def scheduler(self):
        return self._scheduler

    @scheduler.setter
    def scheduler(self, s):
        from .schedulers import TableScheduler
        if not isinstance(s, TableScheduler):
            raise TypeError("scheduler must be blobctrl_amd.schedulers.UniPCMultistepScheduler or DDIMScheduler")
        self._scheduler = s

    def to(self, *a, **k):
        return self

    def set_progress_bar_config(self, **k):
        pass

    # ---- LoRA (D/loaders/lora_pipeline.py:60-117 -> D/loaders/unet.py:271-340): UNet adapters only, merged when the UNet is packed
    def load_lora_weights(self, pretrained_model_name_or_path_or_dict, adapter_name=None, weight_name=None, **kwargs):
        """`pipeline.load_lora_weights(unet_lora_path, adapter_name="default")` (inf:270-273).  A directory (with
        `pytorch_lora_weights.safetensors` or `weight_name`), a .safetensors file, or a state dict; keys under `unet.` go to the
        UNet (lora_pipeline.py:102-110); text-encoder LoRA keys are refused (the released BlobCtrl adapter has none)."""
        import os
        from .checkpoint import load_lora, lora_from_state_dict
        src = pretrained_model_name_or_path_or_dict
        if isinstance(src, dict):
            if any(k.startswith("text_encoder.") for k in src):
                raise NotImplementedError("text-encoder LoRA is not supported")
            lora, alphas = lora_from_state_dict(src)
        else:
            path = os.path.join(src, weight_name) if weight_name and os.path.isdir(src) else src
            lora, alphas = load_lora(path)
        name = adapter_name if adapter_name is not None else f"default_{len(self.unet._adapters)}"
        self.unet.load_lora_adapter(lora, alphas, adapter_name=name)

    def set_adapters(self, adapter_names, adapter_weights=None):
        """`pipeline.set_adapters(["default"])` (inf:274): the active adapters and their weights (merged on the next call)."""
        self.unet.set_adapters(adapter_names, adapter_weights)

    def fuse_lora(self, *a, **k):
        """Adapters are always merged into the packed weights: nothing to do."""

    def unload_lora_weights(self):
        self.unet.unload_lora()

    def get_active_adapters(self):
        return [n for n, a in self.unet._adapters.items() if a["active"]]

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet.config.time_cond_proj_dim is None                # pipe:494-497

    @property
    def clip_skip(self):
        return self._clip_skip

    @property
    def num_timesteps(self):
        return self._num_timesteps

    # ---- pipe:328-436
    def check_inputs(self, prompt, callback_steps, negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None,
                     ip_adapter_image=None, ip_adapter_image_embeds=None, blobnet_conditioning_scale=1.0, control_guidance_start=0.0,
                     control_guidance_end=1.0, callback_on_step_end_tensor_inputs=None):
        if callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if callback_on_step_end_tensor_inputs is not None and not all(k in self._callback_tensor_inputs
                                                                      for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found "
                             f"{[k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `negative_prompt`: {negative_prompt} and `negative_prompt_embeds`:"
                             f" {negative_prompt_embeds}. Please make sure to only forward one of the two.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                             f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")
        if not isinstance(blobnet_conditioning_scale, float):
            raise TypeError("For single blobnet: `blobnet_conditioning_scale` must be type `float`.")
        if not isinstance(control_guidance_start, (tuple, list)):
            control_guidance_start = [control_guidance_start]
        if not isinstance(control_guidance_end, (tuple, list)):
            control_guidance_end = [control_guidance_end]
        if len(control_guidance_start) != len(control_guidance_end):
            raise ValueError(f"`control_guidance_start` has {len(control_guidance_start)} elements, but `control_guidance_end` has "
                             f"{len(control_guidance_end)} elements. Make sure to provide the same number of elements to each list.")
        for start, end in zip(control_guidance_start, control_guidance_end):
            if start >= end:
                raise ValueError(f"control guidance start: {start} cannot be larger or equal to control guidance end: {end}.")
            if start < 0.0:
                raise ValueError(f"control guidance start: {start} can't be smaller than 0.")
            if end > 1.0:
                raise ValueError(f"control guidance end: {end} can't be larger than 1.0.")
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            raise NotImplementedError("IP-Adapter inputs are not part of the BlobCtrl path (the scripts never pass them)")

    # ---- pipe:508-687
    def _tokenize(self, text, max_length):
        ids = self.tokenizer(text, padding="max_length", max_length=max_length, truncation=True, return_tensors="pt").input_ids
        return torch.as_tensor(ids, dtype=torch.int64)

    def encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                      prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                      lora_scale: Optional[float] = None, clip_skip: Optional[int] = None):
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        if prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("this pipeline was built without tokenizer / text_encoder: pass prompt_embeds")
            ids = self._tokenize(prompt, getattr(self.tokenizer, "model_max_length", 77))
            prompt_embeds = self.text_encoder(ids.to(self.device), clip_skip=clip_skip)[0]
        prompt_embeds = prompt_embeds.to(self.device)
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt, seq_len, -1)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if negative_prompt is None:
                uncond_tokens = [""] * batch_size
            elif prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond_tokens = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                                 " the batch size of `prompt`.")
            else:
                uncond_tokens = negative_prompt
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("this pipeline was built without tokenizer / text_encoder: pass negative_prompt_embeds")
            ids = self._tokenize(uncond_tokens, prompt_embeds.shape[1])
            negative_prompt_embeds = self.text_encoder(ids.to(self.device))[0]
        if do_classifier_free_guidance:
            seq_len = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.to(self.device).repeat(1, num_images_per_prompt, 1)
            negative_prompt_embeds = negative_prompt_embeds.view(batch_size * num_images_per_prompt, seq_len, -1)
        return prompt_embeds, negative_prompt_embeds

    # ---- pipe:300-309 (the posterior sample draws from the GLOBAL generator there; `generator` makes it reproducible here)
    def encode_latents(self, image, device=None, dtype=None, prompt_embeds=None, height=512, width=512, generator=None):
        if self.vae is None:
            raise ValueError("this pipeline was built without a VAE: it cannot encode images")
        if not torch.is_tensor(image):
            image = self.image_processor.preprocess(image, height=height, width=width)
        lat = self.engine.encode_latents(image.to(self.device, torch.float32), generator)
        return lat if prompt_embeds is None else lat.repeat(prompt_embeds.shape[0], 1, 1, 1)

    # ---- pipe:690-703
    def encode_image_dinov2(self, image, device=None):
        if self.dinov2 is None:
            raise ValueError("this pipeline was built without dinov2: it cannot embed the foreground image")
        if not torch.is_tensor(image):
            image = self.dinov2_processor.preprocess(images=image, do_resize=True, return_tensors="pt", do_convert_rgb=True)["pixel_values"]
        emb = self.dinov2(pixel_values=torch.as_tensor(image).to(self.device, torch.float32)).pooler_output
        return emb.unsqueeze(1) if emb.ndim == 2 else emb

    # ---- pipe:438-453 + D/utils/torch_utils.py:38-83 (a CPU generator draws on the CPU, then the noise moves to the device)
    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                noise = torch.cat([torch.randn((1,) + shape[1:], generator=g_, device=g_.device, dtype=torch.float32).cpu()
                                   for g_ in generator], 0)
            else:
                gdev = generator.device if generator is not None else "cpu"
                noise = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
        else:
            noise = latents
        noise = noise.to(self.device, torch.float32)
        return noise * self.scheduler.init_noise_sigma, noise

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, fg_image=None, bg_image=None, gs_score: torch.Tensor = None,
                 height: Optional[int] = 512, width: Optional[int] = 512, num_inference_steps: int = 50, timesteps: List[int] = None,
                 guidance_scale: float = 7.5, negative_prompt: Optional[Union[str, List[str]]] = None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 ip_adapter_image=None, ip_adapter_image_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs=None, blobnet_conditioning_scale: Union[float, List[float]] = 1.0,
                 blobnet_control_guidance_start: Union[float, List[float]] = 0.0,
                 blobnet_control_guidance_end: Union[float, List[float]] = 1.0, clip_skip: Optional[int] = None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs: List[str] = ["latents"], return_sample: bool = False,
                 **kwargs):
        """pipeline_blobnet.py:743-1166 with the same keyword names, defaults and error behaviour.  Returns
        StableDiffusionBlobNetPipelineOutput(images=..., nsfw_content_detected=None) or `(images, None)` for return_dict=False;
        `output_type` "pil" | "np" | "pt" | "latent"."""
        callback_steps = kwargs.pop("callback_steps", None)
        if kwargs.pop("callback", None) is not None:
            raise NotImplementedError("`callback` is deprecated in the reference (pipe:868-875): use callback_on_step_end")
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")
        if timesteps is not None:
            raise NotImplementedError("custom `timesteps` are not tabulated; pass num_inference_steps")
        if cross_attention_kwargs:
            raise NotImplementedError("cross_attention_kwargs (runtime LoRA scale) are not supported: LoRA is merged at load")
        # 0.1 align the control-guidance format (pipe:884-893)
        if not isinstance(blobnet_control_guidance_start, list) and isinstance(blobnet_control_guidance_end, list):
            blobnet_control_guidance_start = len(blobnet_control_guidance_end) * [blobnet_control_guidance_start]
        elif not isinstance(blobnet_control_guidance_end, list) and isinstance(blobnet_control_guidance_start, list):
            blobnet_control_guidance_end = len(blobnet_control_guidance_start) * [blobnet_control_guidance_end]
        elif not isinstance(blobnet_control_guidance_start, list) and not isinstance(blobnet_control_guidance_end, list):
            blobnet_control_guidance_start, blobnet_control_guidance_end = [blobnet_control_guidance_start], [blobnet_control_guidance_end]
        # 1. check inputs (pipe:897-909)
        self.check_inputs(prompt, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds, ip_adapter_image,
                          ip_adapter_image_embeds, blobnet_conditioning_scale, blobnet_control_guidance_start,
                          blobnet_control_guidance_end, callback_on_step_end_tensor_inputs)
        if gs_score is None:
            raise ValueError("gs_score is required (pipe:974)")
        # 2. call parameters (pipe:912-922)
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        self._guidance_scale, self._clip_skip = guidance_scale, clip_skip
        cfg = self.do_classifier_free_guidance
        # 3. prompt (pipe:937-949)
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(prompt, self.device, num_images_per_prompt, cfg, negative_prompt,
                                                                   prompt_embeds=prompt_embeds,
                                                                   negative_prompt_embeds=negative_prompt_embeds, clip_skip=clip_skip)
        if cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        # 5./6. timesteps and latents (pipe:953-968); the noise is drawn BEFORE the images are encoded, like the reference
        self.scheduler.set_timesteps(num_inference_steps)
        self._num_timesteps = num_inference_steps
        lat0, noise = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, height, width,
                                           prompt_embeds.dtype, self.device, generator, latents)
        # 7. image latents (pipe:970-971): fg first, then bg - the order in which the reference consumes the global generator
        fg_lat = self.encode_latents(fg_image, height=height, width=width)
        bg_lat = self.encode_latents(bg_image, height=height, width=width)
        # 9. DINOv2 feature of the foreground (pipe:982)
        dino = self.encode_image_dinov2(fg_image)
        # 12. loop (pipe:1025-1123) on the captured plans
        cb = None
        if callback_on_step_end is not None:
            def cb(_engine, i, t, kw):
                out = callback_on_step_end(self, i, t, {k: kw[k] for k in callback_on_step_end_tensor_inputs if k in kw})
                return out
        final = self.engine.denoise(prompt_embeds, fg_lat, bg_lat, gs_score.to(self.device), dino,
                                    num_inference_steps=num_inference_steps, guidance_scale=float(guidance_scale), latents=noise,
                                    blobnet_conditioning_scale=blobnet_conditioning_scale,
                                    blobnet_control_guidance_start=blobnet_control_guidance_start[0],
                                    blobnet_control_guidance_end=blobnet_control_guidance_end[0], output_type="latent",
                                    callback_on_step_end=cb, return_sample=return_sample, eta=eta, do_classifier_free_guidance=cfg)
        # pipe:1132-1166
        if output_type != "latent":
            if self.vae is None:
                raise ValueError("this pipeline was built without a VAE: use output_type='latent'")
            image = self.vae.decode(final / self.vae.config.scaling_factor, return_dict=False)[0]
        else:
            image = final
        image = self.image_processor.postprocess(image, output_type=output_type, do_denormalize=[True] * image.shape[0])
        if not return_dict:
            return (image, None)
        return StableDiffusionBlobNetPipelineOutput(images=image, nsfw_content_detected=None)
