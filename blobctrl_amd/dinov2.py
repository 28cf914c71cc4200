"""DINOv2 (ViT-L/14) pooled image embedding on MI355X: replaces `self.dinov2(**inputs).pooler_output`
(blobctrl/pipelines/pipeline_blobnet.py:690-703; arithmetic of transformers' Dinov2Model, pinned 4.49.0 upstream).

Runs once per edit: patch-embed (im2col + GEMM) -> [CLS] + position embeddings -> L x { x + ls1*Attn(LN x), x + ls2*MLP(LN x) }
-> LayerNorm -> token 0.  Reuses the hot path's LayerNorm / MFMA GEMM / flash-attention kernels; LayerScale, bias, exact-erf
GELU and the residual adds are GEMM epilogues.  Position embeddings are bicubic-resized on the host at pack time when the
input grid differs from the training grid (Dinov2Embeddings.interpolate_pos_encoding).
"""
import torch
import torch.nn.functional as F

from . import _lib
from .launch import Recorder, run_graphed
from .weights import pad8


class Dinov2Model:
    def __init__(self, state_dict, num_heads: int, patch_size: int = 14, eps: float = 1e-6, device="cuda:0"):
        self.device = torch.device(device)
        # (a host device is accepted for CONSTRUCTION only - loading / inspecting checkpoints; running refuses it: `_need_gpu`)
        _lib.load()
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self.sd = sd
        self.heads, self.patch, self.eps = num_heads, patch_size, eps
        self.D = sd["embeddings.cls_token"].shape[-1]
        self.L = 0
        while f"encoder.layer.{self.L}.norm1.weight" in sd:
            self.L += 1
        dev = self.device
        h, f = {}, {}
        w = sd["embeddings.patch_embeddings.projection.weight"].flatten(1)
        self.Kpad = pad8(w.shape[1])
        wp = torch.zeros(w.shape[0], self.Kpad)
        wp[:, : w.shape[1]] = w
        h["patch.weight"] = wp.half().to(dev)
        f["patch.bias"] = sd["embeddings.patch_embeddings.projection.bias"].to(dev)
        f["cls"] = sd["embeddings.cls_token"].reshape(-1).to(dev)
        for i in range(self.L):
            p = f"encoder.layer.{i}."
            a = p + "attention.attention."
            h[p + "qk.weight"] = torch.cat([sd[a + "query.weight"], sd[a + "key.weight"]], 0).half().to(dev)
            f[p + "qk.bias"] = torch.cat([sd[a + "query.bias"], sd[a + "key.bias"]], 0).to(dev)
            h[p + "v.weight"] = sd[a + "value.weight"].half().to(dev)
            f[p + "v.bias"] = sd[a + "value.bias"].to(dev)
            for src, dst in (("attention.output.dense", "o"), ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
                h[p + dst + ".weight"] = sd[p + src + ".weight"].half().to(dev)
                f[p + dst + ".bias"] = sd[p + src + ".bias"].to(dev)
            for nm in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "layer_scale1.lambda1",
                       "layer_scale2.lambda1"):
                f[p + nm] = sd[p + nm].to(dev)
        f["layernorm.weight"] = sd["layernorm.weight"].to(dev)
        f["layernorm.bias"] = sd["layernorm.bias"].to(dev)
        self.h, self.f = h, f
        self._plans = {}

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", **_ignored):
        """`Dinov2Model.from_pretrained(dinov2_path)` (inf:257): config.json + model.safetensors."""
        from .checkpoint import _config, _model_file, read_safetensors
        cfg = _config(path)
        return cls(read_safetensors(_model_file(path)), num_heads=cfg.get("num_attention_heads", 16),
                   patch_size=cfg.get("patch_size", 14), eps=cfg.get("layer_norm_eps", 1e-6), device=device)

    def _pos(self, gh, gw):
        pos = self.sd["embeddings.position_embeddings"]
        n = pos.shape[1] - 1
        if not (n == gh * gw and gh == gw):
            side = int(round(n ** 0.5))
            patch = pos[:, 1:].reshape(1, side, side, self.D).permute(0, 3, 1, 2)
            patch = F.interpolate(patch.float(), size=(gh, gw), mode="bicubic", align_corners=False)
            pos = torch.cat([pos[:, :1], patch.permute(0, 2, 3, 1).reshape(1, -1, self.D)], dim=1)
        return pos.reshape(-1, self.D).contiguous().to(self.device)

    def to(self, *a, **k):
        return self

    def _need_gpu(self):
        if self.device.type != "cuda":
            raise _lib.BlobCtrlHipError("blobctrl_amd.Dinov2Model runs on MI355X only; there is no CPU fallback")

    def _plan(self, B, H, W):
        self._need_gpu()
        key = (B, H, W)
        if key in self._plans:
            return self._plans[key]
        rec = Recorder(self.device)
        P = type("Plan", (), {})()
        P.rec = rec
        D, heads = self.D, self.heads
        d = D // heads
        gh, gw = H // self.patch, W // self.patch
        T = gh * gw
        N = T + 1
        M = B * N
        hw, fw = self.h, self.f
        P.pixels = rec.zeros(B, 3, H, W, dtype=torch.float32)
        P.pos = self._pos(gh, gw)
        P.seg = rec.begin("dinov2")
        cols = rec.empty(B * T, self.Kpad)
        rec.call("bc_patchify", P.pixels.data_ptr(), B, H, W, self.patch, self.Kpad, cols.data_ptr(), kind="patchify")
        pe = rec.empty(B * T, D)
        rec.gemm(A=cols, W=hw["patch.weight"], M=B * T, N=D, K=self.Kpad, out=pe, bias=fw["patch.bias"], kind="patch_embed")
        x = rec.empty(M, D)
        rec.call("bc_add_cls_pos", pe.data_ptr(), fw["cls"].data_ptr(), P.pos.data_ptr(), B, T, D, x.data_ptr(),
                 kind="cls_pos")
        ldvt = (N + 63) // 64 * 64
        for i in range(self.L):
            p = f"encoder.layer.{i}."
            ln = rec.layernorm(x, M, D, fw[p + "norm1.weight"], fw[p + "norm1.bias"], self.eps)
            qk = rec.empty(M, 2 * D)
            rec.gemm(A=ln, W=hw[p + "qk.weight"], M=M, N=2 * D, K=D, out=qk, bias=fw[p + "qk.bias"], kind="qkv")
            vt = rec.zeros(B, D, ldvt)
            rec.gemm(A=ln, W=hw[p + "v.weight"], M=M, N=D, K=D, out=vt, bias=fw[p + "v.bias"], out_mode=_lib.OUT_F16_T,
                     ldc=ldvt, rows_per_batch=N, kind="qkv")
            a = rec.empty(M, D)
            rec.attention(qk, qk, vt, a, B, heads, d, N, N, 2 * D, 2 * D, ldvt, D, N * 2 * D, N * 2 * D, D * ldvt, N * D,
                          d ** -0.5, q_off=0, k_off=D)
            x2 = rec.empty(M, D)
            rec.gemm(A=a, W=hw[p + "o.weight"], M=M, N=D, K=D, out=x2, bias=fw[p + "o.bias"],
                     colscale=fw[p + "layer_scale1.lambda1"], R=x, ldr=D, kind="attn_out")
            ln = rec.layernorm(x2, M, D, fw[p + "norm2.weight"], fw[p + "norm2.bias"], self.eps)
            m1 = rec.empty(M, 4 * D)
            rec.gemm(A=ln, W=hw[p + "fc1.weight"], M=M, N=hw[p + "fc1.weight"].shape[0], K=D, out=m1,
                     bias=fw[p + "fc1.bias"], act=_lib.ACT_GELU, kind="ff")
            x = rec.empty(M, D)
            rec.gemm(A=m1, W=hw[p + "fc2.weight"], M=M, N=D, K=hw[p + "fc2.weight"].shape[1], out=x,
                     bias=fw[p + "fc2.bias"], colscale=fw[p + "layer_scale2.lambda1"], R=x2, ldr=D, kind="ff")
        # final LayerNorm on token 0 of every image only (row stride N*D)
        P.pooled16 = rec.layernorm(x, B, D, fw["layernorm.weight"], fw["layernorm.bias"], self.eps, ldx=N * D, ldy=D)
        self._plans[key] = P
        return P

    @torch.no_grad()
    def pooler_output(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [B,3,H,W] (H, W multiples of the patch size) -> pooled CLS [B, D] fp32."""
        B, C, H, W = pixel_values.shape
        if C != 3 or H % self.patch or W % self.patch:
            raise ValueError("pixel_values must be [B,3,H,W] with H, W multiples of the patch size")
        P = self._plan(B, H, W)
        P.pixels.copy_(pixel_values.to(self.device, torch.float32))
        run_graphed(P.seg, self.device)
        return P.pooled16.float()

    def __call__(self, pixel_values):
        out = type("Dinov2Output", (), {})()
        out.pooler_output = self.pooler_output(pixel_values)
        return out
