"""Weight pipeline: reference-schema state_dict (fp32, CPU) -> LoRA merge -> packed fp16 device layouts.

Layouts (all K-contiguous so that both MFMA operands are read with ds_read_b128):
  * conv3x3  [Co, Ci, 3, 3]  ->  [Co, 9 * Ci_pad]  with k = (ky*3 + kx) * Ci_pad + ci     (NHWC implicit-GEMM order)
  * conv1x1 / Linear [Co, Ci] ->  [Co, Ci_pad]
  * GEGLU proj [8C, C] (first half value, second half gate: D/models/activations.py:117-123) -> rows interleaved in
    groups of 64 = 32 value rows | 32 gate rows so that one wavefront holds matching value/gate accumulators.
  * to_q | to_k concatenated along N (one GEMM), to_v separate (its epilogue writes V transposed for the attention kernel).
  * all ResnetBlock2D.time_emb_proj of a net concatenated along N (one GEMM per step for the 22 projections).
  * biases, norm scales and shifts stay fp32.

LoRA (reference: D/loaders/unet.py:271-340, peft semantics W + (alpha / r) * B @ A; peft itself is absent from the
reference tree, so this merge is pinned only against the closed form - "LoRA parity unpinned").
"""
from collections import OrderedDict
from typing import Dict, Optional

import torch


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def merge_lora(sd: Dict[str, torch.Tensor], lora: Dict[str, torch.Tensor], alphas: Optional[Dict[str, float]] = None,
               adapter_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Return a copy of `sd` with every `<module>.lora_A.weight` / `<module>.lora_B.weight` pair merged into
    `<module>.weight`.  Rank r = lora_B.shape[1]; scale = alpha / r with alpha defaulting to r (D/loaders/unet.py:314-340)."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for key in list(lora.keys()):
        if not key.endswith(".lora_A.weight"):
            continue
        mod = key[: -len(".lora_A.weight")]
        a = lora[key].float()
        b = lora[mod + ".lora_B.weight"].float()
        r = b.shape[1]
        alpha = float(alphas.get(mod, r)) if alphas else float(r)
        w = out[mod + ".weight"]
        delta = (b.flatten(1) @ a.flatten(1)).reshape(w.shape)
        out[mod + ".weight"] = w + (adapter_scale * alpha / r) * delta
    return out


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    co, ci = w.shape[:2]
    cip = pad8(ci)
    p = torch.zeros(co, 3, 3, cip, dtype=torch.float32)
    p[..., :ci] = w.permute(0, 2, 3, 1)
    return p.reshape(co, 9 * cip)


WREG_TILES = ((0, 3), (3, 2), (5, 2), (7, 3))      # (first column tile, tiles) of the four column groups of csrc/conv_wreg.hip


def pack_conv_wreg(w: torch.Tensor) -> torch.Tensor:
    """Packed 3x3 weights [N][9 * Cin] (k = (ky * 3 + kx) * Cin + c: `pack_conv3x3`) -> the per-wave fragment streams of BC_TILE_WREG
    (csrc/conv_wreg.hip, same as `bc_conv_wreg_pack`): per 160-column block, per (column group, K half of the 64-channel chunk) one
    contiguous stream [chunk][kx][ky][tile][lane = 16 (k sub-chunk) + row][8]."""
    N, K = w.shape
    Cin = K // 9
    assert N % 160 == 0 and Cin % 64 == 0 and K == 9 * Cin, (N, K)
    nch = Cin // 64
    # [block, tile 10, row 16, ky, kx, chunk, kg, q 4, 8]
    v = w.reshape(N // 160, 10, 16, 3, 3, nch, 2, 4, 8)
    parts = []
    for blk in range(N // 160):
        for t0, nt in WREG_TILES:
            g = v[blk, t0:t0 + nt]                                   # [t, row, ky, kx, chunk, kg, q, 8]
            g = g.permute(5, 4, 3, 2, 0, 6, 1, 7)                     # [kg, chunk, kx, ky, t, q, row, 8]
            parts.append(g.reshape(-1))
    return torch.cat(parts).reshape(N, K)


def pack_matrix(w: torch.Tensor) -> torch.Tensor:
    w = w.reshape(w.shape[0], -1)
    co, ci = w.shape
    cip = pad8(ci)
    if cip == ci:
        return w.contiguous()
    p = torch.zeros(co, cip, dtype=torch.float32)
    p[:, :ci] = w
    return p


def interleave_geglu(w: torch.Tensor, b: torch.Tensor):
    """[8C, C] / [8C] -> rows ordered (32 value, 32 gate) per group."""
    n = w.shape[0] // 2
    assert n % 32 == 0, "GEGLU inner dim must be a multiple of 32"
    wv, wg = w[:n].reshape(n // 32, 32, -1), w[n:].reshape(n // 32, 32, -1)
    bv, bg = b[:n].reshape(n // 32, 32), b[n:].reshape(n // 32, 32)
    return torch.cat([wv, wg], dim=1).reshape(2 * n, -1).contiguous(), torch.cat([bv, bg], dim=1).reshape(2 * n).contiguous()


def pack_gemm_wreg(w: torch.Tensor, nt: int) -> torch.Tensor:
    """[N][K] matrix -> the per-wave fragment streams of a BC_TILE_GW* configuration with `nt` 16-column tiles per wave
    (csrc/gemm_wreg.hip, same bytes as `bc_gemm_wreg_pack`): [column tile][wave 4][k-step][tile][lane = 16 q + r][8], lane l holding
    w[n0 + 16 tile + (l & 15)][32 s + 8 (l >> 4) : + 8]; followed by the 32 fragments the register ring reads past the end (zeros)."""
    N, K = w.shape
    assert N % (64 * nt) == 0 and K % 32 == 0, (N, K, nt)
    v = w.reshape(N // (64 * nt), 4, nt, 16, K // 32, 4, 8)            # [j, wave, t, r, s, q, e]
    v = v.permute(0, 1, 4, 2, 5, 3, 6).reshape(-1)                      # [j, wave, s, t, q, r, e]
    return torch.cat([v, torch.zeros(32 * 512, dtype=w.dtype, device=w.device)])


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm -> Linear folded for BcGemm.ln_colsum (csrc/gemm_wreg.hip): returns (W' = fp16(W diag(gamma)), colsum of the ROUNDED
    W' in fp32, bias' = bias + W beta in fp32), so that Linear(LN(x)) = rstd (x W'^T - mean colsum) + bias'."""
    wf = w.float()
    w2 = (wf * gamma.float()[None, :]).half()
    cs = w2.float().sum(1)
    b2 = wf @ beta.float()
    if bias is not None:
        b2 = b2 + bias.float()
    return w2, cs.contiguous(), b2.contiguous()


def fold_cross_attention(k: torch.Tensor, vt: torch.Tensor, T: int, heads: int, scale: float, wq: torch.Tensor, bq: torch.Tensor, wo: torch.Tensor):
    """Host statement of bc_ctx_fold (csrc/gemm_wreg.hip): k [B][T][C] projected context rows, vt [B][C][>= T] its V^T, (wq, bq, wo) =
    PackedTrunk.ctx_fold_weights.  Returns (QK [B][128 heads][C] fp16 - row 128 h + j = key j of head h, zero from j = T on -, colsum of the
    rounded rows fp32, bias fp32 [B][128 heads], VO [B][C][80 heads] fp16) as plain matrices - pack_gemm_wreg(QK[b], 2) /
    pack_gemm_wreg(VO[b], 2) are the streams the kernel writes."""
    B, _, C = k.shape
    D = C // heads
    kh = torch.zeros(B, heads, 128, D, dtype=torch.float32, device=k.device)
    kh[:, :, :T] = k[:, :T].float().reshape(B, T, heads, D).permute(0, 2, 1, 3)
    qk = (torch.einsum("bhjd,hdc->bhjc", kh, wq.float().reshape(heads, D, C)) * scale).half().reshape(B, heads * 128, C)
    qb = (torch.einsum("bhjd,hd->bhj", kh, bq.float().reshape(heads, D)) * scale).reshape(B, heads * 128)
    vh = torch.zeros(B, heads, D, 80, dtype=torch.float32, device=k.device)
    vh[..., :T] = vt[:, :, :T].float().reshape(B, heads, D, T)
    vo = torch.einsum("nhd,bhdj->bnhj", wo.float().reshape(C, heads, D), vh).half().reshape(B, C, heads * 80)
    return qk, qk.float().sum(-1), qb, vo


class PackedTrunk:
    """Device-resident packed weights of one trunk (UNet or BlobNet)."""

    def gw(self, key: str, tile_cfg: int, ln: Optional[str] = None, bias: Optional[str] = None, extra=()):
        """Fragment stream of the matrix `key` (or of the row-concatenation key + extra) for a BC_TILE_GW* configuration, made on first
        use.  `ln` = name prefix of a LayerNorm folded into it (gamma into the weights, beta into the bias, column sums for the
        kernel's mean correction): returns (stream, colsum or None, bias tensor or None)."""
        from . import _lib
        nt = _lib.GW_TILES[tile_cfg]
        ck = ("gw", key, tuple(extra), nt, ln)
        cache = self.__dict__.setdefault("_gw", {})
        if ck not in cache:
            w = self.h[key] if not extra else torch.cat([self.h[key]] + [self.h[e] for e in extra], 0)
            b = self.f[bias] if bias else None
            cs = None
            if ln is not None:
                w, cs, b = fold_layernorm(w, b, self.f[ln + ".weight"], self.f[ln + ".bias"])
            cache[ck] = (pack_gemm_wreg(w, nt), cs, b)
        return cache[ck]

    def ff2_proj_out(self, p: str) -> str:
        """`ff.net.2` and `proj_out` of the Transformer2D block `p` as ONE weight (round 5): proj_out(ff.net.2(g) + b2 + h2) + bpo =
        [P F2 | P] . [g | h2] + (P b2 + bpo) - the two projections are consecutive linear maps with only the residual add between them
        (attention.py:530-541 `ff_output + hidden_states`, transformer_2d.py:516-523 `proj_out`), so the product is made once, in
        fp32, when the weights are packed, and the step runs one two-source GEMM (K = 4C + C) instead of two dependent launches.
        Returns the key of the merged matrix [C][5C] (its bias under key + ".bias")."""
        k = p + "ff2_proj_out"
        if k + ".weight" not in self.h:
            bp = p + "transformer_blocks.0."
            P = self.h[p + "proj_out.weight"].float()
            F2 = self.h[bp + "ff.net.2.weight"].float()
            self.h[k + ".weight"] = torch.cat([P @ F2, P], 1).half().contiguous()
            self.f[k + ".bias"] = (P @ self.f[bp + "ff.net.2.bias"].float() + self.f[p + "proj_out.bias"].float()).contiguous()
        return k

    def qkv_weight(self, bp: str) -> str:
        """Key of attn1's to_q | to_k | to_v as ONE row-concatenated matrix [3C][C] (round 6: one BC_TILE_G256 launch writes q | k row-major
        and V^T through BcGemm.C_t; attention_processor.py:2191-2196)."""
        k = bp + "attn1.to_qkv.weight"
        if k not in self.h:
            self.h[k] = torch.cat([self.h[bp + "attn1.to_qk.weight"], self.h[bp + "attn1.to_v.weight"]], 0).contiguous()
        return k

    def ctx_fold_weights(self, bp: str):
        """What bc_ctx_fold multiplies the projected prompt with (block prefix `bp` = "...transformer_blocks.0."): (wq = fp16 of
        attn2.to_q.weight diag(gamma of norm2) [C][C], bq = to_q.weight . beta of norm2 fp32 [C], wo = attn2.to_out.0.weight [C][C] as stored) -
        the LayerNorm in front of to_q folded as for BcGemm.ln_colsum (fold_layernorm)."""
        cache = self.__dict__.setdefault("_ctx_fold", {})
        if bp not in cache:
            wq, _, bq = fold_layernorm(self.h[bp + "attn2.to_q.weight"], None, self.f[bp + "norm2.weight"], self.f[bp + "norm2.bias"])
            cache[bp] = (wq.contiguous(), bq)
        return cache[bp] + (self.h[bp + "attn2.to_out.0.weight"],)

    def wreg(self, key: str) -> str:
        """Key of the BC_TILE_WREG fragment stream of the packed 3x3 weight `key` (made on first use, kept beside the matrix)."""
        k2 = key + "_wreg"
        if k2 not in self.h:
            self.h[k2] = pack_conv_wreg(self.h[key])
        return k2

    def __init__(self, sd: Dict[str, torch.Tensor], device, block_out_channels, layers_per_block=2):
        self.device = device
        self.h: Dict[str, torch.Tensor] = {}       # fp16 matrices
        self.f: Dict[str, torch.Tensor] = {}       # fp32 vectors
        self.boc = tuple(block_out_channels)
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.has_cross = any(".attn2." in k for k in sd)
        temb_w, temb_b, self.temb_slices = [], [], {}
        off = 0
        for k, v in sd.items():
            if k.endswith(".bias") or (v.ndim == 1):
                if ".time_emb_proj." in k or ".ff.net.0.proj." in k:
                    continue
                self.f[k] = v.to(device)
                continue
            if ".time_emb_proj.weight" in k:
                p = k[: -len("time_emb_proj.weight")]
                temb_w.append(v)
                temb_b.append(sd[p + "time_emb_proj.bias"])
                self.temb_slices[p] = (off, v.shape[0])
                off += v.shape[0]
            elif ".ff.net.0.proj.weight" in k:
                w, b = interleave_geglu(v, sd[k[:-6] + "bias"])
                self.h[k] = w.half().to(device)
                self.f[k[:-6] + "bias"] = b.to(device)
            elif k.endswith("attn1.to_q.weight"):
                p = k[: -len("to_q.weight")]
                self.h[p + "to_qk.weight"] = torch.cat([v, sd[p + "to_k.weight"]], 0).half().to(device)
            elif k.endswith("attn1.to_k.weight"):
                continue
            elif v.ndim == 4 and v.shape[-1] == 3:
                self.h[k] = pack_conv3x3(v).half().to(device)
            else:
                self.h[k] = pack_matrix(v).half().to(device)
        self.temb_total = off
        w_in = sd["conv_in.weight"]
        co = w_in.shape[0]

        def k128(w8ch):
            """[co][3][3][8] -> [co][128]: the dense-GEMM form of an 8-channel conv_in (k = tap * 8 + channel, zero from k = 72),
            multiplied with the im2col operand bc_assemble_input_im2col writes."""
            out = torch.zeros(co, 128)
            out[:, :72] = w8ch.reshape(co, 72)
            return out
        if w_in.shape[1] <= 8:
            base = torch.zeros(co, 3, 3, 8)
            base[..., : w_in.shape[1]] = w_in.permute(0, 2, 3, 1)
            self.h["conv_in.weight_k128"] = k128(base).half().to(device)
        if w_in.shape[1] > 8:
            # BlobNet: conditioning = 1 score channel + F feature channels that are (score x per-edit vector) (pipe:706-721).
            # Keep the F-channel block as a matrix whose per-edit GEMV (engine.record_collapse) collapses it into ONE extra input
            # channel (channel 5) of an 8-channel conv_in.  Row co * 16 + tap of the matrix (taps 9..15: zero rows), so that the
            # GEMV's output row m lands at element m * 8 + 5 of the [co][128] dense-GEMM weight.
            base = torch.zeros(co, 3, 3, 8)
            base[..., :5] = w_in[:, :5].permute(0, 2, 3, 1)
            self.h["conv_in.weight8"] = k128(base).half().to(device)
            fm = torch.zeros(co, 16, w_in.shape[1] - 5)
            fm[:, :9] = w_in[:, 5:].permute(0, 2, 3, 1).reshape(co, 9, -1)
            self.h["conv_in.featmat"] = pack_matrix(fm.reshape(co * 16, -1)).half().to(device)
        self.h["temb_all.weight"] = torch.cat(temb_w, 0).half().to(device)
        self.f["temb_all.bias"] = torch.cat(temb_b, 0).to(device)
        self._to_arenas()

    # ---- arenas: all fp16 matrices live in ONE contiguous buffer, all fp32 vectors in another, so that a replica is
    # ---- two RCCL broadcasts (blobctrl_amd/dist.py) and one allocation each; views are 256-byte aligned.
    @staticmethod
    def _layout(tensors, align_elems):
        layout, off = [], 0
        for k, t in tensors.items():
            layout.append((k, tuple(t.shape), off))
            off += (t.numel() + align_elems - 1) // align_elems * align_elems
        return layout, off

    def _to_arenas(self):
        self.h_layout, hn = self._layout(self.h, 128)
        self.f_layout, fn = self._layout(self.f, 64)
        self.h_arena = torch.zeros(hn, dtype=torch.float16, device=self.device)
        self.f_arena = torch.zeros(fn, dtype=torch.float32, device=self.device)
        self._view(copy_from=(self.h, self.f))

    def _view(self, copy_from=None):
        h, f = {}, {}
        for (k, shape, off) in self.h_layout:
            n = 1
            for d in shape:
                n *= d
            h[k] = self.h_arena[off:off + n].view(shape)
            if copy_from is not None:
                h[k].copy_(copy_from[0][k])
        for (k, shape, off) in self.f_layout:
            n = 1
            for d in shape:
                n *= d
            f[k] = self.f_arena[off:off + n].view(shape)
            if copy_from is not None:
                f[k].copy_(copy_from[1][k])
        self.h, self.f = h, f

    def meta(self):
        """Everything a replica needs besides the two arenas (small, picklable)."""
        return dict(h_layout=self.h_layout, f_layout=self.f_layout, temb_slices=self.temb_slices,
                    temb_total=self.temb_total, has_cross=self.has_cross, boc=self.boc,
                    h_elems=self.h_arena.numel(), f_elems=self.f_arena.numel())

    @classmethod
    def from_meta(cls, meta, device):
        """An EMPTY replica with the same layout (filled by an RCCL broadcast of the arenas)."""
        self = cls.__new__(cls)
        self.device = device
        self.h_layout, self.f_layout = meta["h_layout"], meta["f_layout"]
        self.temb_slices, self.temb_total = meta["temb_slices"], meta["temb_total"]
        self.has_cross, self.boc = meta["has_cross"], meta["boc"]
        self.h_arena = torch.zeros(meta["h_elems"], dtype=torch.float16, device=device)
        self.f_arena = torch.zeros(meta["f_elems"], dtype=torch.float32, device=device)
        self._view()
        return self

    def derived_nbytes(self):
        """Bytes of the fragment streams made on first use BESIDE the arenas (conv_wreg / gemm_wreg / row-chain streams: re-ordered
        copies of matrices the arenas also hold in row-major form for the kernels that take that form - other batch sizes, canvases
        the halo kernels do not cover, the diagnostic switches).  Not broadcast: every rank derives its own."""
        n = sum(t.numel() * t.element_size() for k, t in self.h.items() if k.endswith("_wreg"))
        for cache in (self.__dict__.get("_gw", {}), self.__dict__.get("_rowchain", {}), self.__dict__.get("_ctx_fold", {})):
            for v in cache.values():
                n += sum(t.numel() * t.element_size() for t in v if torch.is_tensor(t))
        return n

    def nbytes(self):
        """Device bytes of this trunk: the two arenas plus the derived fragment streams made so far."""
        return self.h_arena.numel() * 2 + self.f_arena.numel() * 4 + self.derived_nbytes()


# ---------------------------------------------------------------------------------------------------------------- row-chain streams
RC_HC, RC_RPAD = 128, 20                     # mirrors csrc/rowchain.hip (RC_HC, RC_RPAD)
RC_CHANNELS = (320, 640)                     # block widths rowchain.hip is instantiated for (C / 80 waves per 64-row workgroup)


def _frag_stream(w: torch.Tensor, row_starts, k0: int = 0, k1: int = None) -> torch.Tensor:
    """[N][K] fp16 matrix -> per-wave MFMA fragment stream [waves][KS * NT][64 lanes][8] for one GEMM segment of rowchain.hip.
    `row_starts[wave][tile]` = first of the 16 output rows of that wave's tile; lane l of a fragment holds
    W[row_start + (l & 15)][k0 + 32 s + 8 (l >> 4) : + 8] (the v_mfma_f32_16x16x32_f16 operand layout), k-step s outermost."""
    k1 = w.shape[1] if k1 is None else k1
    rs = torch.as_tensor(row_starts, device=w.device)                       # [waves][NT]
    nw, nt = rs.shape
    rows = rs[:, :, None] + torch.arange(16, device=w.device)               # [waves][NT][16]
    x = w[rows.reshape(-1), k0:k1].reshape(nw, nt, 16, (k1 - k0) // 32, 4, 8)  # [wave][tile][m][s][q][8]
    return x.permute(0, 3, 1, 4, 2, 5).reshape(nw, -1, 64, 8).contiguous()  # [wave][s][tile][q][m][8] -> lane = 16 q + m


def pack_rowchain_kv(k: torch.Tensor, vt: torch.Tensor, T: int, pad_frags: int = 20) -> torch.Tensor:
    """Host statement of bc_rowchain_pack_kv (csrc/rowchain.hip): the projected context of one block - K rows `k` [B][T][C], V^T `vt`
    [B][C][>= T] - as the per-(image, wave) fragment streams BC_CHAIN_MIDX consumes: [B][C / 80 waves][fragments + pad_frags][64 lanes][8]
    fp16.  8 heads; a wave's 80 channels are 2 heads of 40 (C = 320) or one of 80 (C = 640).  Per head: K as [k-step][5 key tiles] - lane
    16 q + m holds key 16 t + m, channels 32 (s0 + ks) + 8 q .. + 7 of the operand image's k-step (zero outside the head's channels and
    for keys >= T) - then V^T as [key k-step 0..2][value tile] - lane holds channel 80 w + 16 (tlo + tv) + m, keys 32 ks + 8 q .. + 7."""
    B, _, C = k.shape
    D = C // 8
    hpw, qks, vtiles = 80 // D, (2 if D == 40 else 3), (3 if D == 40 else 5)
    fr_head = 5 * qks + 3 * vtiles
    nw = C // 80
    out = torch.zeros(B, nw, hpw * fr_head + pad_frags, 64, 8, dtype=torch.float16, device=k.device)
    kp = torch.zeros(B, 80, C + 64, dtype=torch.float16, device=k.device)      # keys padded to 80, channels padded for the last k-step
    kp[:, :T, :C] = k[:, :T]
    vp = torch.zeros(B, C, 96, dtype=torch.float16, device=k.device)
    vp[:, :, :T] = vt[:, :, :T]
    lane = torch.arange(64, device=k.device)
    m, q = lane % 16, lane // 16
    j = torch.arange(8, device=k.device)
    for w in range(nw):
        for hh in range(hpw):
            c_h = 80 * w + D * hh
            s0 = c_h // 32
            f0 = hh * fr_head
            for ks in range(qks):
                for t in range(5):
                    ch = 32 * (s0 + ks) + 8 * q[:, None] + j[None, :]                  # [64][8]
                    val = kp[:, (16 * t + m)[:, None].expand(64, 8), ch]              # [B][64][8]
                    inside = (ch >= c_h) & (ch < c_h + D)
                    out[:, w, f0 + ks * 5 + t] = torch.where(inside[None], val, torch.zeros_like(val))
            tlo = 2 if (D == 40 and hh == 1) else 0
            for ks in range(3):
                for tv in range(vtiles):
                    chv = 80 * w + 16 * (tlo + tv) + m                                 # [64]
                    key = 32 * ks + 8 * q[:, None] + j[None, :]
                    val = vp[:, chv[:, None].expand(64, 8), key]
                    inside = ((chv >= c_h) & (chv < c_h + D))[:, None].expand(64, 8)
                    out[:, w, f0 + 5 * qks + ks * vtiles + tv] = torch.where(inside[None], val, torch.zeros_like(val))
    return out


def pack_rowchain(pw: "PackedTrunk", p: str, kind: int, zero_name: Optional[str] = None, nsplit: int = 1):
    """Weights of one 320- or 640-channel Transformer2D block `p` (e.g. "down_blocks.0.attentions.0.") as rowchain.hip consumes them:
    (wstream [C / 80 waves][fragments][64][8] fp16 in exact consumption order + RC_RPAD fragments of padding, vec fp32).
    kind 0 IN: proj_in, to_q, to_k, to_v | vec = proj_in bias, norm1 gamma, norm1 beta
    kind 1 MID: attn1.to_out, attn2.to_q | vec = to_out bias, norm2 gamma, beta
    kind 2 OUT: attnX.to_out, per 128-wide hidden chunk (ff.net.0.proj (value, gate) tile pairs, ff.net.2 K-slice), proj_out[, zero-conv]
                | vec = to_out bias, norm3 gamma, beta, GEGLU bias per (chunk, wave, pass: value 16 | gate 16), ff.net.2 bias, proj_out bias
                [, zero-conv bias]
    kind 3 OUT_FF (nsplit slices): per slice z: attnX.to_out, the hidden chunks [z, z + 1) * (4C / 128) / nsplit -> wstream
                [nsplit][waves][fragments][64][8]; kind 4 OUT_TAIL: proj_out[, zero-conv].  Both take the vec of kind 2.
    kind 6 OUT_FFP (nsplit slices): as kind 3 with proj_out[, zero-conv] behind every slice's chunks (the reduction over the slices
                happens behind them: bc_rowchain_sum); vec of kind 2.
    Built on the device from the trunk's packed matrices (every rank builds its own: nothing to broadcast)."""
    h, f = pw.h, pw.f
    bp = p + "transformer_blocks.0."
    C = h[p + "proj_in.weight"].shape[0]
    assert C in RC_CHANNELS, C
    nw = C // 80
    nC = [[80 * w + 16 * t for t in range(5)] for w in range(nw)]
    segs, vec = [], []
    if kind == 0:
        qk = h[bp + "attn1.to_qk.weight"]
        segs = [_frag_stream(h[p + "proj_in.weight"], nC), _frag_stream(qk[:C], nC), _frag_stream(qk[C:], nC),
                _frag_stream(h[bp + "attn1.to_v.weight"], nC)]
        vec = [f[p + "proj_in.bias"], f[bp + "norm1.weight"], f[bp + "norm1.bias"]]
    elif kind == 1:
        segs = [_frag_stream(h[bp + "attn1.to_out.0.weight"], nC), _frag_stream(h[bp + "attn2.to_q.weight"], nC)]
        vec = [f[bp + "attn1.to_out.0.bias"], f[bp + "norm2.weight"], f[bp + "norm2.bias"]]
    else:
        att = "attn2" if pw.has_cross else "attn1"
        w1i, b1i = h[bp + "ff.net.0.proj.weight"], f[bp + "ff.net.0.proj.bias"]     # rows interleaved (32 value | 32 gate) per group
        hid = 4 * C
        w1 = w1i.reshape(hid // 32, 2, 32, C)
        b1 = b1i.reshape(hid // 32, 2, 32)
        wv, wg = w1[:, 0].reshape(hid, C), w1[:, 1].reshape(hid, C)                # original order: value rows / gate rows
        bv, bg = b1[:, 0].reshape(hid), b1[:, 1].reshape(hid)
        w1o = torch.cat([wv, wg], 0)
        w2 = h[bp + "ff.net.2.weight"]
        seg_to_out = _frag_stream(h[bp + att + ".to_out.0.weight"], nC)
        ffb, chunks = [], []
        hpw = RC_HC // nw                         # hidden units of a chunk per wave (32 / 16)
        for c in range(hid // RC_HC):
            cs = []
            for tp in range(hpw // 16):           # (value tile, gate tile) passes over K per chunk
                cs.append(_frag_stream(w1o, [[RC_HC * c + hpw * w + 16 * tp, hid + RC_HC * c + hpw * w + 16 * tp] for w in range(nw)]))
            cs.append(_frag_stream(w2, nC, RC_HC * c, RC_HC * (c + 1)))
            chunks.append(cs)
            for w in range(nw):
                for tp in range(hpw // 16):
                    j0 = RC_HC * c + hpw * w + 16 * tp
                    ffb += [bv[j0:j0 + 16], bg[j0:j0 + 16]]
        seg_proj_out = _frag_stream(h[p + "proj_out.weight"], nC)
        if kind == 2:
            segs = [seg_to_out] + [x for cs in chunks for x in cs] + [seg_proj_out]
        elif kind == 4:
            segs = [seg_proj_out]
        vec = [f[bp + att + ".to_out.0.bias"], f[bp + "norm3.weight"], f[bp + "norm3.bias"], torch.cat(ffb), f[bp + "ff.net.2.bias"],
               f[p + "proj_out.bias"]]
        seg_zero = None
        if zero_name is not None:
            seg_zero = _frag_stream(h[zero_name + ".weight"], nC)
            if kind not in (3, 6):
                segs.append(seg_zero)
            vec.append(f[zero_name + ".bias"])
        if kind in (3, 6):
            nchunks = len(chunks)
            assert nchunks % nsplit == 0, (nchunks, nsplit)
            pad = torch.zeros(nw, RC_RPAD, 64, 8, dtype=torch.float16, device=seg_to_out.device)
            per = nchunks // nsplit
            tail = ([seg_proj_out] + ([seg_zero] if seg_zero is not None else [])) if kind == 6 else []
            slices = [torch.cat([seg_to_out] + [x for cs in chunks[z * per:(z + 1) * per] for x in cs] + tail + [pad], 1) for z in range(nsplit)]
            return torch.stack(slices).contiguous(), torch.cat([v.reshape(-1).float() for v in vec]).contiguous()
    pad = torch.zeros(nw, RC_RPAD, 64, 8, dtype=torch.float16, device=segs[0].device)
    return torch.cat(segs + [pad], 1).contiguous(), torch.cat([v.reshape(-1).float() for v in vec]).contiguous()
