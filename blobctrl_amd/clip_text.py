"""CLIP text encoder (SD-1.5: 12 layers, 768-d, 12 heads, 77 tokens) on MI355X: replaces
`self.text_encoder(text_input_ids)[0]` of blobctrl/pipelines/pipeline_blobnet.py:599-611 (arithmetic of transformers'
CLIPTextModel, pinned 4.49.0 upstream), including the clip_skip branch.  Runs once per edit.

Token + position embedding gather -> L x { x + out_proj(causal attention(LN1 x)) ; x + fc2(quick_gelu(fc1(LN2 x))) } -> final
LayerNorm.  Reuses the hot path's LayerNorm / MFMA GEMM / flash-attention kernels (causal-mask entry point); biases, quick_gelu
and the residual adds are GEMM epilogues.  Tokenisation stays on the host (callers pass `input_ids`).
"""
import torch

from . import _lib
from .launch import Recorder, run_graphed


class CLIPTextModel:
    def __init__(self, state_dict, num_heads: int = 12, eps: float = 1e-5, device="cuda:0"):
        self.device = torch.device(device)
        # (a host device is accepted for CONSTRUCTION only - loading / inspecting checkpoints; running refuses it: `_need_gpu`)
        _lib.load()
        # on-disk layout has a "text_model." prefix (transformers 4.x); 5.x dropped it
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v.detach().float().cpu()
              for k, v in state_dict.items()}
        dev = self.device
        self.heads, self.eps = num_heads, eps
        self.vocab, self.D = sd["embeddings.token_embedding.weight"].shape
        self.max_pos = sd["embeddings.position_embedding.weight"].shape[0]
        if self.D % num_heads or (self.D // num_heads) not in (8, 16, 32, 40, 64, 80, 160):
            raise ValueError(f"unsupported head_dim {self.D}/{num_heads}")
        self.L = 0
        while f"encoder.layers.{self.L}.layer_norm1.weight" in sd:
            self.L += 1
        h, f = {}, {}
        h["tok"] = sd["embeddings.token_embedding.weight"].half().to(dev)
        f["pos"] = sd["embeddings.position_embedding.weight"].to(dev)
        for i in range(self.L):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            h[p + "qk.weight"] = torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"]], 0).half().to(dev)
            f[p + "qk.bias"] = torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"]], 0).to(dev)
            for src, dst in ((a + "v_proj", "v"), (a + "out_proj", "o"), (p + "mlp.fc1", "fc1"), (p + "mlp.fc2", "fc2")):
                h[p + dst + ".weight"] = sd[src + ".weight"].half().to(dev)
                f[p + dst + ".bias"] = sd[src + ".bias"].to(dev)
            for nm in ("layer_norm1.weight", "layer_norm1.bias", "layer_norm2.weight", "layer_norm2.bias"):
                f[p + nm] = sd[p + nm].to(dev)
        f["final.weight"] = sd["final_layer_norm.weight"].to(dev)
        f["final.bias"] = sd["final_layer_norm.bias"].to(dev)
        self.h, self.f = h, f
        self._plans = {}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, device="cuda:0", **_ignored):
        """`CLIPTextModel.from_pretrained(sd15_path, subfolder="text_encoder")`: config.json + model.safetensors."""
        import os
        from .checkpoint import _config, _model_file, read_safetensors
        d = os.path.join(path, subfolder) if subfolder else path
        cfg = _config(d)
        return cls(read_safetensors(_model_file(d)), num_heads=cfg.get("num_attention_heads", 12),
                   eps=cfg.get("layer_norm_eps", 1e-5), device=device)

    def to(self, *a, **k):
        return self

    def _need_gpu(self):
        if self.device.type != "cuda":
            raise _lib.BlobCtrlHipError("blobctrl_amd.CLIPTextModel runs on MI355X only; there is no CPU fallback")

    def _plan(self, B, T):
        self._need_gpu()
        key = (B, T)
        if key in self._plans:
            return self._plans[key]
        rec = Recorder(self.device)
        P = type("Plan", (), {})()
        P.rec = rec
        D, heads = self.D, self.heads
        d = D // heads
        M = B * T
        hw, fw = self.h, self.f
        P.ids = torch.zeros(B, T, dtype=torch.int64, device=self.device)
        P.seg = rec.begin("clip_text")
        x = rec.empty(M, D)
        rec.call("bc_embed_tokens", P.ids.data_ptr(), hw["tok"].data_ptr(), fw["pos"].data_ptr(), B, T, D, self.vocab,
                 x.data_ptr(), kind="embed", keep=(P.ids, x))
        ldvt = (T + 63) // 64 * 64
        P.hidden = [x]
        for i in range(self.L):
            p = f"encoder.layers.{i}."
            ln = rec.layernorm(x, M, D, fw[p + "layer_norm1.weight"], fw[p + "layer_norm1.bias"], self.eps)
            qk = rec.empty(M, 2 * D)
            rec.gemm(A=ln, W=hw[p + "qk.weight"], M=M, N=2 * D, K=D, out=qk, bias=fw[p + "qk.bias"], kind="qkv")
            vt = rec.zeros(B, D, ldvt)
            rec.gemm(A=ln, W=hw[p + "v.weight"], M=M, N=D, K=D, out=vt, bias=fw[p + "v.bias"], out_mode=_lib.OUT_F16_T,
                     ldc=ldvt, rows_per_batch=T, kind="qkv")
            a = rec.empty(M, D)
            rec.attention(qk, qk, vt, a, B, heads, d, T, T, 2 * D, 2 * D, ldvt, D, T * 2 * D, T * 2 * D, D * ldvt, T * D,
                          d ** -0.5, q_off=0, k_off=D, causal=True)
            x2 = rec.empty(M, D)
            rec.gemm(A=a, W=hw[p + "o.weight"], M=M, N=D, K=D, out=x2, bias=fw[p + "o.bias"], R=x, ldr=D, kind="attn_out")
            ln = rec.layernorm(x2, M, D, fw[p + "layer_norm2.weight"], fw[p + "layer_norm2.bias"], self.eps)
            F1 = hw[p + "fc1.weight"].shape[0]
            m1 = rec.empty(M, F1)
            rec.gemm(A=ln, W=hw[p + "fc1.weight"], M=M, N=F1, K=D, out=m1, bias=fw[p + "fc1.bias"], act=_lib.ACT_QUICK_GELU,
                     kind="ff")
            x = rec.empty(M, D)
            rec.gemm(A=m1, W=hw[p + "fc2.weight"], M=M, N=D, K=F1, out=x, bias=fw[p + "fc2.bias"], R=x2, ldr=D, kind="ff")
            P.hidden.append(x)
        P.final = {}
        for skip, src in ((None, P.hidden[-1]),) + tuple((s, P.hidden[-(s + 1)]) for s in range(1, min(self.L, 4))):
            P.final[skip] = rec.layernorm(src, M, D, fw["final.weight"], fw["final.bias"], self.eps)
        self._plans[key] = P
        return P

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, attention_mask=None, clip_skip=None):
        """input_ids [B, T] int64 (T <= max_position_embeddings) -> (prompt_embeds [B, T, D] fp16,), as `text_encoder(ids)[0]`;
        `clip_skip=k` returns final_layer_norm(hidden_states[-(k+1)]) (pipe:604-611)."""
        if attention_mask is not None:
            raise NotImplementedError("SD-1.5's text encoder config has no use_attention_mask (pipe:593-596)")
        B, T = input_ids.shape
        if T > self.max_pos:
            raise ValueError(f"sequence length {T} exceeds max_position_embeddings {self.max_pos}")
        P = self._plan(B, T)
        if clip_skip not in P.final:
            raise ValueError(f"clip_skip={clip_skip} not supported (0 < clip_skip < min(layers, 4))")
        P.ids.copy_(input_ids.to(self.device, torch.int64))
        run_graphed(P.seg, self.device)
        return (P.final[clip_skip].view(B, T, self.D).clone(),)
