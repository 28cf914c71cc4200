"""SD-1.5 AutoencoderKL (encode / decode) on MI355X — the stages either side of the denoise loop.

Keeps the reference call surface used by the pipeline:
    vae.encode(image).latent_dist.sample(generator)          (pipeline_blobnet.py:300-309; the reference forgets to pass the
                                                              generator there - `sample()` here takes one)
    vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]      (pipeline_blobnet.py:1133)
Host mirror of D/models/autoencoders/autoencoder_kl.py:249-326 and D/models/autoencoders/vae.py:47-348 on the hot path's
kernels: convs / 1x1 / linears = bc_gemm (fused nearest-2x upsample, the encoder's pad-(0,1,0,1) stride-2 downsample via
`conv_nopad_lo`), GroupNorm(eps 1e-6)+SiLU with statistics from the producer epilogues, and the mid block's single-head,
head_dim-512 attention as GEMM (QK^T, scale folded) -> bc_softmax_rows -> GEMM (P V) per image (V produced transposed).
"""
from typing import Optional

import torch

from . import _lib
from .launch import Recorder, run_graphed
from .weights import pack_conv3x3, pack_matrix, pad8


class _Act:
    __slots__ = ("t", "C", "H", "W")

    def __init__(self, t, C, H, W):
        self.t, self.C, self.H, self.W = t, C, H, W


class _Dist:
    """`latent_dist` of the reference's AutoencoderKLOutput (vae.py:767-789)."""

    def __init__(self, vae, moments, B, h, w):
        self._vae, self._mom, self._B, self._h, self._w = vae, moments, B, h, w

    def sample(self, generator: Optional[torch.Generator] = None, scale: float = 1.0) -> torch.Tensor:
        v = self._vae
        Cz = v.latent_channels
        dev = v.device
        noise = torch.randn((self._B, Cz, self._h, self._w), generator=generator,
                            device=generator.device if generator is not None else "cpu", dtype=torch.float32).to(dev)
        out = torch.empty(self._B, Cz, self._h, self._w, dtype=torch.float32, device=dev)
        _lib.check(v.lib.bc_gaussian_sample(self._mom.data_ptr(), noise.data_ptr(), self._B, Cz, self._h * self._w, scale,
                                            out.data_ptr(), torch.cuda.current_stream().cuda_stream), "bc_gaussian_sample")
        return out

    def mode(self) -> torch.Tensor:
        m = self._mom.view(self._B, self._h, self._w, -1)[..., : self._vae.latent_channels]
        return m.permute(0, 3, 1, 2).float().contiguous()

    @property
    def parameters(self):
        return self._mom.view(self._B, self._h, self._w, -1).permute(0, 3, 1, 2).float().contiguous()


class AutoencoderKL:
    class _Cfg:
        scaling_factor = 0.18215

    def __init__(self, state_dict, norm_num_groups: int = 32, layers_per_block: int = 2, device="cuda:0"):
        self.device = torch.device(device)
        # (a host device is accepted for CONSTRUCTION only - loading / inspecting checkpoints; running refuses it: `_need_gpu`)
        self.lib = _lib.load()
        self.config = self._Cfg()
        self.G, self.lpb = norm_num_groups, layers_per_block
        self.config.block_out_channels = None       # (set below from the weights)
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.latent_channels = sd["post_quant_conv.weight"].shape[0]
        self.h, self.f = {}, {}
        dev = self.device
        for k, v in sd.items():
            if v.ndim == 1:
                self.f[k] = v.to(dev)
            elif v.ndim == 4 and v.shape[-1] == 3:
                self.h[k] = pack_conv3x3(v).half().to(dev)
            else:
                self.h[k] = pack_matrix(v).half().to(dev)
        # q | k fused along N for the mid-block attention
        for side in ("encoder", "decoder"):
            a = f"{side}.mid_block.attentions.0."
            self.h[a + "to_qk.weight"] = torch.cat([sd[a + "to_q.weight"], sd[a + "to_k.weight"]], 0).half().to(dev)
            self.f[a + "to_qk.bias"] = torch.cat([sd[a + "to_q.bias"], sd[a + "to_k.bias"]], 0).to(dev)
        i = 0
        boc = []
        while f"encoder.down_blocks.{i}.resnets.0.conv1.weight" in sd:
            boc.append(sd[f"encoder.down_blocks.{i}.resnets.0.conv1.weight"].shape[0])
            i += 1
        self.config.block_out_channels = tuple(boc)
        self._plans = {}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, device="cuda:0", **_ignored):
        """`AutoencoderKL.from_pretrained(sd15_path, subfolder="vae")`: config.json + diffusion_pytorch_model.safetensors."""
        import os
        from .checkpoint import _config, _model_file, read_safetensors
        d = os.path.join(path, subfolder) if subfolder else path
        cfg = _config(d)
        vae = cls(read_safetensors(_model_file(d)), norm_num_groups=cfg.get("norm_num_groups", 32),
                  layers_per_block=cfg.get("layers_per_block", 2), device=device)
        vae.config.scaling_factor = cfg.get("scaling_factor", 0.18215)
        return vae

    # ------------------------------------------------------------------------------------------------ recorded blocks
    def _conv(self, rec, B, x: _Act, name, Cout, stride=1, up=False, nopad_lo=False, R=None, want_gn=True, out_f32=False):
        Hv, Wv = (2 * x.H, 2 * x.W) if up else (x.H, x.W)
        pads = 1 if nopad_lo else 2
        Ho, Wo = (Hv + pads - 3) // stride + 1, (Wv + pads - 3) // stride + 1
        out = rec.empty(B, Ho * Wo, Cout, dtype=torch.float32 if out_f32 else torch.float16)
        kw = {}
        if R is not None:
            kw.update(R=R.t, ldr=R.C)
        rec.gemm(A=x.t, W=self.h[name + ".weight"], M=B * Ho * Wo, N=Cout, K=9 * x.C, out=out,
                 out_mode=_lib.OUT_F32 if out_f32 else _lib.OUT_F16,
                 conv=dict(Cin=x.C, Hin=x.H, Win=x.W, Hv=Hv, Wv=Wv, Hout=Ho, Wout=Wo, stride=stride, nopad_lo=nopad_lo),
                 bias=self.f[name + ".bias"], rows_per_batch=Ho * Wo, want_gn=want_gn and not out_f32, kind="vae_conv", **kw)
        return _Act(out, Cout, Ho, Wo)

    def _dense(self, rec, x_t, M, K, wkey, bkey, N, rows_per_batch=0, **kw):
        out = kw.pop("out", None)
        if out is None:
            out = rec.empty(M, N)
        rec.gemm(A=x_t, W=self.h[wkey], M=M, N=N, K=K, out=out, bias=self.f[bkey] if bkey else None,
                 rows_per_batch=rows_per_batch, kind="vae_dense", **kw)
        return out

    def _gn(self, rec, B, x: _Act, name, silu=True):
        out = rec.groupnorm(x.t, x.C, None, 0, B, x.H * x.W, self.G, 1e-6, self.f[name + ".weight"], self.f[name + ".bias"], silu)
        return _Act(out, x.C, x.H, x.W)

    def _res(self, rec, B, p, x: _Act, Cout):
        h = self._gn(rec, B, x, p + "norm1")
        h = self._conv(rec, B, h, p + "conv1", Cout)
        h = self._gn(rec, B, h, p + "norm2")
        if (p + "conv_shortcut.weight") in self.h:
            M = B * x.H * x.W
            sc = _Act(self._dense(rec, x.t, M, x.C, p + "conv_shortcut.weight", p + "conv_shortcut.bias", Cout), Cout, x.H, x.W)
        else:
            sc = x
        return self._conv(rec, B, h, p + "conv2", Cout, R=sc)

    def _mid(self, rec, B, p, x: _Act):
        x = self._res(rec, B, p + "resnets.0.", x, x.C)
        a = p + "attentions.0."
        Cc, N = x.C, x.H * x.W
        if N % 8:
            raise ValueError("VAE attention needs a token count that is a multiple of 8")
        M = B * N
        n = self._gn(rec, B, x, a + "group_norm", silu=False)
        qk = self._dense(rec, n.t, M, Cc, a + "to_qk.weight", a + "to_qk.bias", 2 * Cc)
        vt = rec.zeros(B, Cc, pad8(N))
        self._dense(rec, n.t, M, Cc, a + "to_v.weight", a + "to_v.bias", Cc, rows_per_batch=N, out=vt,
                    out_mode=_lib.OUT_F16_T, ldc=pad8(N))
        o = rec.empty(M, Cc)
        Np = pad8(N)
        s = rec.zeros(N, Np)                               # scores of ONE image (images run back to back on the stream)
        for b in range(B):                                 # heads = 1, head_dim = C: plain GEMMs per image
            rec.gemm(A=qk, a_offset=b * N * 2 * Cc, lda=2 * Cc, W=qk, w_offset=b * N * 2 * Cc + Cc, M=N, N=N, K=Cc, out=s,
                     ldc=Np, alpha=Cc ** -0.5, kind="vae_attn", ldw=2 * Cc)
            rec.call("bc_softmax_rows", s.data_ptr(), N, N, Np, kind="softmax", keep=(s,))
            rec.gemm(A=s, lda=Np, W=vt, w_offset=b * Cc * Np, M=N, N=Cc, K=Np, out=o, out_offset=b * N * Cc, kind="vae_attn",
                     ldw=Np)
        out = self._dense(rec, o, M, Cc, a + "to_out.0.weight", a + "to_out.0.bias", Cc, rows_per_batch=N, R=x.t, ldr=Cc,
                          want_gn=True)
        return self._res(rec, B, p + "resnets.1.", _Act(out, Cc, x.H, x.W), Cc)

    # ------------------------------------------------------------------------------------------------ plans
    def to(self, *a, **k):
        return self

    def _need_gpu(self):
        if self.device.type != "cuda":
            raise _lib.BlobCtrlHipError("blobctrl_amd.AutoencoderKL runs on MI355X only; there is no CPU fallback")

    def _plan_decode(self, B, h, w):
        self._need_gpu()
        key = ("dec", B, h, w)
        if key in self._plans:
            return self._plans[key]
        rec = Recorder(self.device)
        P = type("Plan", (), {})()
        P.rec = rec
        Cz = self.latent_channels
        P.z = rec.zeros(B, h * w, 8)                                    # latent channels padded to 8
        P.seg = rec.begin("vae_decode")
        zp = rec.zeros(B, h * w, 8)                                     # post-quant output, again padded to 8 channels
        self._dense(rec, P.z, B * h * w, 8, "post_quant_conv.weight", "post_quant_conv.bias", Cz, out=zp, ldc=8)
        x = _Act(zp, 8, h, w)
        rev = [self.h[f"decoder.up_blocks.{i}.resnets.0.conv1.weight"].shape[0] for i in range(4)]
        x = self._conv(rec, B, x, "decoder.conv_in", rev[0])
        x = self._mid(rec, B, "decoder.mid_block.", x)
        i = 0
        while f"decoder.up_blocks.{i}.resnets.0.conv1.weight" in self.h:
            for j in range(self.lpb + 1):
                x = self._res(rec, B, f"decoder.up_blocks.{i}.resnets.{j}.", x, rev[i])
            if f"decoder.up_blocks.{i}.upsamplers.0.conv.weight" in self.h:
                x = self._conv(rec, B, x, f"decoder.up_blocks.{i}.upsamplers.0.conv", rev[i], up=True)
            i += 1
        x = self._gn(rec, B, x, "decoder.conv_norm_out")
        P.img = self._conv(rec, B, x, "decoder.conv_out", self.h["decoder.conv_out.weight"].shape[0], out_f32=True)
        self._plans[key] = P
        return P

    def _plan_encode(self, B, H, W):
        self._need_gpu()
        key = ("enc", B, H, W)
        if key in self._plans:
            return self._plans[key]
        rec = Recorder(self.device)
        P = type("Plan", (), {})()
        P.rec = rec
        P.x = rec.zeros(B, H * W, 8)                                    # RGB padded to 8 channels
        P.seg = rec.begin("vae_encode")
        x = _Act(P.x, 8, H, W)
        boc = [self.h[f"encoder.down_blocks.{i}.resnets.0.conv1.weight"].shape[0] for i in range(4)]
        x = self._conv(rec, B, x, "encoder.conv_in", boc[0])
        i = 0
        while f"encoder.down_blocks.{i}.resnets.0.conv1.weight" in self.h:
            for j in range(self.lpb):
                x = self._res(rec, B, f"encoder.down_blocks.{i}.resnets.{j}.", x, boc[i])
            if f"encoder.down_blocks.{i}.downsamplers.0.conv.weight" in self.h:
                x = self._conv(rec, B, x, f"encoder.down_blocks.{i}.downsamplers.0.conv", boc[i], stride=2, nopad_lo=True)
            i += 1
        x = self._mid(rec, B, "encoder.mid_block.", x)
        x = self._gn(rec, B, x, "encoder.conv_norm_out")
        Cm = 2 * self.latent_channels
        x = self._conv(rec, B, x, "encoder.conv_out", Cm, want_gn=False)
        P.moments = self._dense(rec, x.t, B * x.H * x.W, Cm, "quant_conv.weight", "quant_conv.bias", Cm)
        P.hw = (x.H, x.W)
        self._plans[key] = P
        return P

    # ------------------------------------------------------------------------------------------------ API
    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = False, generator=None):
        """z [B, Cz, h, w] (already divided by the scaling factor) -> (image [B, 3, 8h, 8w] fp32,)."""
        B, Cz, h, w = z.shape
        if Cz != self.latent_channels:
            raise ValueError(f"expected {self.latent_channels} latent channels, got {Cz}")
        P = self._plan_decode(B, h, w)
        s = torch.cuda.current_stream().cuda_stream
        zz = z.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.bc_nchw_to_nhwc_f16(zz.data_ptr(), 1, B, Cz, h * w, 8, P.z.data_ptr(), s), "bc_nchw_to_nhwc_f16")
        run_graphed(P.seg, self.device)
        img = P.img
        out = img.t.view(B, img.H, img.W, img.C).permute(0, 3, 1, 2).contiguous()
        return (out,)

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        """x [B, 3, H, W] in [-1, 1] -> object with `.latent_dist` (sample(generator) / mode() / parameters)."""
        B, C, H, W = x.shape
        if C != 3 or H % 8 or W % 8:
            raise ValueError("image must be [B,3,H,W] with H, W multiples of 8")
        P = self._plan_encode(B, H, W)
        s = torch.cuda.current_stream().cuda_stream
        xx = x.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.bc_nchw_to_nhwc_f16(xx.data_ptr(), 1, B, 3, H * W, 8, P.x.data_ptr(), s), "bc_nchw_to_nhwc_f16")
        run_graphed(P.seg, self.device)
        out = type("AutoencoderKLOutput", (), {})()
        out.latent_dist = _Dist(self, P.moments, B, P.hw[0], P.hw[1])
        return out
