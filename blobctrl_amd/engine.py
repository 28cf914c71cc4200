"""Static launch plans for the two trunks of the hot path: the patched SD-1.5 UNet and BlobNet.

Host-side mirror of (paths relative to the reference root, D/ = diffusers/src/diffusers/):
  * D/models/unets/unet_2d_condition.py:1039-1353  UNet2DConditionModel.forward with down/mid/up_block_add_samples
  * blobctrl/models/blobnet.py:720-945             BlobNetModel.forward
  * D/models/unets/unet_2d_blocks.py:1241-1323, 1378-1433, 860-899, 2514-2624, 2677-2765 (patched blocks)
  * D/models/resnet.py:320-373, transformers/transformer_2d.py:479-527, attention.py:421-541
re-expressed as a flat list of C-ABI launches over NHWC fp16 buffers (see launch.Recorder).  Fusions relative to
the reference op list: channel-concat folded into the GroupNorm / shortcut loaders, nearest-upsample folded into the
following conv's gather, bias / time-embedding / residual / BlobNet right-half add / conditioning scale / GEGLU folded
into GEMM epilogues, V produced transposed by its projection GEMM, q|k projections fused, all 22 time_emb_proj fused.
"""
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import _lib
from .launch import Recorder
from .options import opt
from .weights import PackedTrunk, pad8


@dataclass
class TrunkConfig:
    in_channels: int                      # real conv_in channels (5 for the UNet, 4+1+F for BlobNet)
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_heads: int = 8
    norm_num_groups: int = 32
    cross_attention_dim: Optional[int] = 768
    out_channels: int = 4
    is_blobnet: bool = False


class Act:
    """A token-major NHWC activation [B][H*W][C] (fp16)."""
    __slots__ = ("t", "C", "H", "W")

    def __init__(self, t, C, H, W):
        self.t, self.C, self.H, self.W = t, C, H, W


@dataclass
class Residuals:
    """BlobNet residual tensors in full-canvas token-major layout and how the UNet adds them
    (unet_2d_blocks.py:1303-1307: whole tensor on a square canvas, right-hand square otherwise)."""
    down: List[torch.Tensor]
    mid: torch.Tensor
    up: List[torch.Tensor]
    bmod: int
    events: Optional[dict] = None        # data_ptr -> event signalled by the producer (two-stream execution)


class TrunkPlan:
    """Records one forward of a trunk into the recorder's current segment."""

    def __init__(self, rec: Recorder, pw: PackedTrunk, cfg: TrunkConfig, B: int, H: int, W: int):
        self.rec, self.pw, self.cfg, self.B, self.H, self.W = rec, pw, cfg, B, H, W
        self.G = cfg.norm_num_groups
        self.heads = cfg.num_heads

    # ------------------------------------------------------------------------------------------- helpers
    def _r2(self, res_t, H, W):
        """kwargs for the BlobNet residual add on an H x W output."""
        if res_t is None:
            return {}
        if self.res_events is not None:
            ev = self.res_events.get(res_t.data_ptr())
            if ev is not None:
                self.rec.wait(ev)         # the BlobNet branch (side stream) has produced this residual
        xmin = 0 if W == H else W - H
        return dict(R2=res_t, ldr2=res_t.shape[-1], r2_xmin=xmin, r2_bmod=self.res_bmod, out_w=W)

    def halo_ok(self, Cin, C1, Cout, H, W):
        """The LDS-resident input-halo convolution with the fused GroupNorm prologue (conv_halo.hip) can run this layer."""
        return opt("halo") and bool(self.rec.lib.bc_conv_halo_eligible(Cin, C1, Cout, H, W, H, W, 1))

    def conv3x3(self, x: Act, wname, Cout, stride=1, up_to=None, rowvec=None, R=None, r2=None, out_f32=False,
                kind="conv3x3", out=None, x2: Optional[Act] = None, affine=None, halo=False, tile_cfg=0):
        """`halo=True`: conv_halo.hip on the channel-concat (x | x2) with `affine` = (ab tensor, activation) applied while staging."""
        rec, pw = self.rec, self.pw
        Hv, Wv = up_to if up_to is not None else (x.H, x.W)
        Hout, Wout = (Hv + 2 - 3) // stride + 1, (Wv + 2 - 3) // stride + 1
        M = self.B * Hout * Wout
        if out is None:
            out = rec.empty(self.B, Hout * Wout, Cout, dtype=torch.float32 if out_f32 else torch.float16)
        kw = {}
        if rowvec is not None:
            kw.update(rowvec=rowvec[0], ld_rowvec=rowvec[1])
            if len(rowvec) == 4:                       # per-edit table of all steps: (device step counter, elements per step)
                kw.update(rowvec_idx=rowvec[2], rowvec_step=rowvec[3])
        if R is not None:
            kw.update(R=R.t, ldr=R.C)
        kw.update(self._r2(r2, Hout, Wout))
        Cin = x.C + (x2.C if x2 is not None else 0)
        wkey = wname + ".weight"
        if halo:
            wreg = opt("wreg")                                       # conv_wreg.hip (weights streamed into VGPRs) instead of conv_halo.hip
            kw.update(tile_cfg=_lib.TILE_WREG if wreg else _lib.TILE_HALO, lda=x.C)
            if wreg:
                wkey = pw.wreg(wname + ".weight")
            if x2 is not None:
                kw.update(A2=x2.t, C1=x.C, lda2=x2.C)
            if affine is not None:
                kw.update(a_act=affine[1], **({"a_gn": affine[0]} if isinstance(affine[0], dict) else {"a_affine": affine[0]}))
        else:
            assert x2 is None and affine is None
            # the exact 2x nearest upsample in front of a plain convolution also runs on conv_wreg.hip (source pixel = halo pixel / 2)
            ups_wreg = up_to is not None and (Hv, Wv) == (2 * x.H, 2 * x.W) and stride == 1 and Cin % 64 == 0 and Cout % 160 == 0 and \
                Wv % 16 == 0 and Hv % 8 == 0 and opt("wreg")
            if ups_wreg:
                kw.update(tile_cfg=_lib.TILE_WREG, lda=x.C)
                wkey = pw.wreg(wname + ".weight")
            elif tile_cfg:
                kw.update(tile_cfg=tile_cfg)
        rec.gemm(A=x.t, W=pw.h[wkey], M=M, N=Cout, K=9 * Cin, out=out,
                 out_mode=_lib.OUT_F32 if out_f32 else _lib.OUT_F16,
                 conv=dict(Cin=Cin, Hin=x.H, Win=x.W, Hv=Hv, Wv=Wv, Hout=Hout, Wout=Wout, stride=stride),
                 bias=pw.f[wname + ".bias"], rows_per_batch=Hout * Wout, kind=kind, want_gn=not out_f32, **kw)
        return Act(out, Cout, Hout, Wout)

    def g256_tile(self, M, N, K, kw):
        """BC_TILE_G256 (csrc/gemm256.hip: 256 x 256 tiles, 8 waves, 8-phase LDS-DMA pipeline, persistent workgroups) for a dense
        single-source projection with enough tiles, else 0 = the planner's LDS-DMA tiles.  Stand-alone on an MI355X (tools/g256_probe.py,
        TFLOP/s with the fused epilogue, gemm_fast's best tile in brackets): [8192 x 10240 x 1280] 1233 (673), with GEGLU 1137,
        [8192 x 1280 x 5120] 971 (571), [8192 x 3840 x 1280] 1144, [8192 x 1280 x 1280] 799 (~450; 160 tiles on 256 CUs),
        [16384 x 5120 x 640] 1042.  The tile-count threshold (BC_PLAN g256_min_tiles, default 64): at 80 tiles - BlobNet's and the UNet's
        N = 1280 projections at batch 8 - a launch fills 80 CUs for about the time the 160 half-size tiles take, and leaves the other
        queue the rest (batch 8: 48.39 / 48.30 -> 48.14 ms per step on one box; below 64 tiles it loses)."""
        if not opt("g256") or kw.get("tile_cfg") or kw.get("splitk") not in (None, 1):
            return 0
        if M % 256 or N % 256 or (M // 256) * (N // 256) < opt("g256_min_tiles"):
            return 0
        want_gn = 1 if kw.get("want_gn") else 0
        c1 = kw.get("C1", 0) if kw.get("A2") is not None else 0
        if kw.get("C_t") is not None and (kw.get("n_t0", 0) % 256 or kw.get("ldc_t", 0) % 8):
            return 0
        ok = self.rec.lib.bc_gemm256_eligible(M, N, K, c1, kw.get("out_mode", _lib.OUT_F16), kw.get("rows_per_batch", 0) or M, want_gn)
        return _lib.TILE_G256 if ok else 0

    def dense(self, x_t, M, K, wname, N, bias=True, out=None, kind="linear", wkey=None, **kw):
        rec, pw = self.rec, self.pw
        act = kw.get("act", _lib.ACT_NONE)
        n_out = N // 2 if act == _lib.ACT_GEGLU else N
        if out is None:
            out = rec.empty(M, n_out)
        g256 = self.g256_tile(M, N, K, kw)
        if g256:
            kw["tile_cfg"] = g256
        rec.gemm(A=x_t, W=pw.h[wkey or (wname + ".weight")], M=M, N=N, K=K, out=out,
                 bias=pw.f[wname + ".bias"] if bias else None, kind=kind, **kw)
        return out

    def gn_conv(self, x: Act, x2: Optional[Act], norm, eps, wname, Cout, **kw):
        """GroupNorm(x | x2) -> SiLU -> conv3x3 (resnet.py:327-341, 351-366).  Fused (statistics -> per-channel affine; normalise +
        SiLU inside the convolution's halo staging) when the layer fits conv_halo.hip, else GroupNorm pass + implicit GEMM."""
        pw = self.pw
        C2 = x2.C if x2 is not None else 0
        if self.halo_ok(x.C + C2, C2 and x.C, Cout, x.H, x.W):
            requests = self.B if self.cfg.is_blobnet else max(1, self.B // 2)         # (the UNet runs the CFG pair of every request)
            if opt("gn_pass_min_requests") and requests >= opt("gn_pass_min_requests") and x.C + C2 >= opt("gn_pass_min_cin") and opt("wreg"):
                # batches of >= 4 requests: a GroupNorm-apply pass + the PLAIN conv_wreg kernel.  The fused prologue costs every workgroup of
                # every round ~10k cycles (statistics table, in-place pass over the halo rows: stand-alone 230 vs 195 + 25 us at 640 channels
                # and batch 8, tools/conv_probe.py); the pass costs a launch on the queue, which is what decides at batch 1-2 whatever the grid
                # (same box, ms per step, fused / pass: batch 2 14.38 / 14.50-14.57, batch 4 25.38 / 25.19, batch 8 48.19 / 47.49, 768^2 x 4 74.49 / 73.43)
                return self.conv3x3(self.groupnorm(x, x2, norm, eps, True), wname, Cout, halo=True, **kw)
            gn = dict(x1=x.t, C1=x.C, x2=x2.t if x2 is not None else None, C2=C2, B=self.B, HW=x.H * x.W, G=self.G, eps=eps,
                      gamma=pw.f[norm + ".weight"], beta=pw.f[norm + ".bias"])
            return self.conv3x3(x, wname, Cout, x2=x2, affine=(gn, _lib.ACT_SILU), halo=True, **kw)
        return self.conv3x3(self.groupnorm(x, x2, norm, eps, True), wname, Cout, **kw)

    def groupnorm(self, x: Act, x2: Optional[Act], name, eps, silu):
        pw = self.pw
        C2 = x2.C if x2 is not None else 0
        out = self.rec.groupnorm(x.t, x.C, x2.t if x2 is not None else None, C2, self.B, x.H * x.W, self.G, eps,
                                 pw.f[name + ".weight"], pw.f[name + ".bias"], silu)
        return Act(out, x.C + C2, x.H, x.W)

    # ------------------------------------------------------------------------------------------- blocks
    def resnet(self, p, x: Act, skip: Optional[Act], Cout, r2=None, out=None):
        """resnet.py:320-373 on (x [, skip]) = torch.cat([x, skip], 1).  `out`: the caller's output buffer."""
        pw = self.pw
        Cin = x.C + (skip.C if skip is not None else 0)
        off, n = pw.temb_slices[p]
        assert n == Cout
        rowvec = (self.tproj.data_ptr() + off * 2, pw.temb_total) + getattr(self, "tproj_table", ())
        h = self.gn_conv(x, skip, p + "norm1", 1e-5, p + "conv1", Cout, rowvec=rowvec)
        M = self.B * x.H * x.W
        if (p + "conv_shortcut.weight") in pw.h:
            kw = {}
            if skip is not None:
                kw.update(A2=skip.t, C1=x.C, lda=x.C, lda2=skip.C)
            gw = self.gw_tile(M, Cout, Cin, x.C if skip is not None else 0) if Cout >= 1280 else 0
            if gw:                                        # low-resolution levels: weights streamed into VGPRs (gemm_wreg.hip)
                w, _, b = pw.gw(p + "conv_shortcut.weight", gw, bias=p + "conv_shortcut.bias")
                out = self.rec.empty(M, Cout)
                self.rec.gemm(A=x.t, W=w, M=M, N=Cout, K=Cin, out=out, bias=b, tile_cfg=gw, kind="conv1x1", **kw)
                sc = Act(out, Cout, x.H, x.W)
            else:
                sc = Act(self.dense(x.t, M, Cin, p + "conv_shortcut", Cout, kind="conv1x1", **kw), Cout, x.H, x.W)
        else:
            assert skip is None and Cin == Cout
            sc = x
        return self.gn_conv(h, None, p + "norm2", 1e-5, p + "conv2", Cout, R=sc, r2=r2, **({"out": out} if out is not None else {}))

    def rowchain_ok(self, Cc, M, HW, p=""):
        """The fused row-chain kernels (csrc/rowchain.hip) take this block: 320 or 640 channels, 64-row blocks inside one image, and
        enough row blocks for the price of the structure - every 64-row workgroup streams the block's WHOLE weight set (4.1 MB at 320
        channels, 16.4 MB at 640) at the per-CU fetch rate.  At 320 channels that pays from the first row block.  At 640 channels a
        workgroup needs ~234 us whatever the batch; measured in the step at batch 1 (same box, two rounds): the UNet's 32 x 64 level
        (64 row blocks on 64 CUs, the other 192 left to the BlobNet branch) 10.485 -> 10.29 ms, BlobNet's too (32 row blocks) 10.40
        - hence from 64 row blocks upwards.  Exception: BlobNet's UP blocks.  The UNet queue is the step's critical path and BlobNet
        runs 1.1 - 1.6 ms ahead of it by the time it reaches its 640-channel up blocks (tools/critical_path.py), so there the
        32-workgroup form is the better neighbour even though it is the slower kernel: 10.43 -> 10.385 ms (same box, two rounds)."""
        if not opt("rowchain") or not self.rec.lib.bc_rowchain_supported(Cc, M, HW):
            return False
        min_blocks = {320: 1, 640: opt("rowchain_min_blocks_640")}[Cc]
        if Cc == 640 and self.cfg.is_blobnet and p.startswith("up_blocks"):
            min_blocks = opt("rowchain_min_blocks_640_blob_up")
        return M // 64 >= min_blocks

    def rowchain_ff_split(self, Cc, M):
        """Workgroups per 64-row block for the feed-forward of the row-chain's block end (1 = the one-launch form).  A row block's
        feed-forward is MFMA-bound on ONE CU (640 channels: 120 of the launch's 150 us); with 64 row blocks (UNet, 32 x 64 level) or 32
        (BlobNet) three quarters of the chip idle meanwhile.  BC_PLAN ff_split_640 / ff_split_320 override (must divide 20 / 10 chunks)."""
        blocks = M // 64
        if Cc == 640:
            # (round 5, OUT_FFP: proj_out [+ zero-conv] run inside every slice, so the split no longer pays for itself with a tail
            #  launch that re-reads nsplit fp32 slabs: as many slices as it takes to put a workgroup on every CU - 4 at 64 row blocks,
            #  the UNet at batch 1; BlobNet's 32 row blocks likewise 4 = 128 workgroups)
            dflt = 4 if self.rowchain_ffp() else 2
            if self.cfg.is_blobnet and opt("ff_split_640_blob"):
                return opt("ff_split_640_blob") if blocks <= 64 else 1
            return (opt("ff_split_640") or dflt) if blocks <= 64 else 1
        return opt("ff_split_320") if blocks <= 128 else 1

    def rowchain_ffp(self):
        """The split block end as OUT_FFP + sum (round 5) instead of OUT_FF + OUT_TAIL (round 3)?  OUT_FFP repeats to_out, proj_out [and
        the zero-conv] in every slice: 104 instead of 145 us per block on the UNet's queue - the step's critical path - for ~35 % more
        CU-time.  The step is bound by the CU-time of BOTH queues' kernels at least as much as by the UNet chain (tools/
        concurrent_timeline.py): measured on one box, the UNet takes it (9.27 = 9.27 ms per step at batch 1), BlobNet, which
        runs ahead of the UNet with a millisecond of slack, keeps the form with the smaller footprint (batch 2, where only BlobNet's
        64 row blocks split: 15.73 vs 15.88 ms).  BC_PLAN ffp=0 / ffp_blob=1 override."""
        if not opt("ffp"):
            return False
        return not self.cfg.is_blobnet or opt("ffp_blob")

    def transformer_rowchain(self, p, x: Act, r2=None, zero=None, fan_out=None):
        """One Transformer2D block as 2 (BlobNet) or 3 (UNet) row-chain launches + its attention calls: GroupNorm affine -> proj_in ->
        LayerNorm -> q | k | V^T ; attention ; [to_out + residual -> LayerNorm -> attn2.to_q ; cross-attention ;] to_out + residual ->
        LayerNorm -> GEGLU feed-forward -> proj_out + x (+ BlobNet residual) [-> zero-conv].  `zero` = (name, alpha, alpha_dev,
        alpha_idx, alpha_bstride): BlobNet's zero-conv of the block output, returned as second value.
        `fan_out` = (x_full, extra) - the CFG-invariant prefix (record_forward): self.B is HALF the UNet batch and x lives in the first
        half of x_full [2 self.B][HW][C]; everything up to the self-attention runs on that half, one bc_dup_halves fans h0, the attention
        output, x and `extra` (a list of (tensor, bytes of one half)) out to the image pairs, and the block goes on at 2 self.B from
        the launch that first sees the prompt (attn2: attention.py:504-510)."""
        from .weights import pack_rowchain
        rec, pw, B = self.rec, self.pw, self.B
        Cc, HW = x.C, x.H * x.W
        M = B * HW
        d = Cc // self.heads
        scale = d ** -0.5
        bp = p + "transformer_blocks.0."
        cache = pw.__dict__.setdefault("_rowchain", {})

        def packed(kind, zname=None, nsplit=1):
            key = (p, kind, zname, nsplit)
            if key not in cache:
                cache[key] = pack_rowchain(pw, p, kind, zname, nsplit)
            return cache[key]
        if opt("gn_finalize_launch"):
            gnkw = dict(affine=rec.gn_affine(x.t, Cc, None, 0, B, HW, self.G, 1e-6, pw.f[p + "norm.weight"], pw.f[p + "norm.bias"]))
        else:                                              # the GroupNorm finalize runs in the IN launch's prologue, from the statistics totals
            gnkw = dict(gn_in=(rec.gn_sources(x.t, Cc, None, 0, B, HW, self.G)[0], pw.f[p + "norm.weight"], pw.f[p + "norm.bias"], self.G, 1e-6))
        ldvt = (HW + 63) // 64 * 64
        vt = rec.zeros(B, Cc, ldvt)
        # (the block's head on the column-tiled gemm_wreg projections - all CUs - instead of one 64-row workgroup per row block was
        #  measured at the UNet's 640-channel level: 9.25 vs 9.13 ms per step, slower - the 64-CU launch leaves the chip to BlobNet)
        mult = 2 if fan_out is not None else 1
        h0_all, a_all = rec.empty(mult * M, Cc), rec.empty(mult * M, Cc)
        h0, qk = h0_all[:M], rec.empty(M, 2 * Cc)
        w, v = packed(_lib.CHAIN_IN)
        rec.rowchain(_lib.CHAIN_IN, Cc, M, HW, x.t, w, v, h0, out1=qk, out2=vt, ldvt=ldvt, **gnkw)
        a = a_all[:M]
        rec.attention(qk, qk, vt, a, B, self.heads, d, HW, HW, 2 * Cc, 2 * Cc, ldvt, Cc, HW * 2 * Cc, HW * 2 * Cc, Cc * ldvt, HW * Cc,
                      scale, q_off=0, k_off=Cc)
        if fan_out is not None:
            x_full, extra = fan_out
            half = M * Cc * 2
            rec.dup_halves([(h0_all, half), (a_all, half), (x_full, half)] + list(extra))
            B, M = 2 * B, 2 * M
            h0, a, x = h0_all, a_all, Act(x_full, Cc, x.H, x.W)
        h = h0
        if pw.has_cross:
            h1 = rec.empty(M, Cc)
            w, v = packed(_lib.CHAIN_MID)
            ck, cvt, T, ldc_vt = self.ctx_kv[bp]
            kvs = getattr(self, "ctx_kvs", {}).get(bp)
            if kvs is not None:
                # the cross-attention runs inside the MID launch: q never leaves the workgroup, K / V^T come as fragment streams
                a2 = rec.empty(M, Cc)
                rec.rowchain_midx(Cc, M, HW, a, w, v, kvs, T, self.heads, scale, h1, a2, res=h0)
                a = a2
            else:
                q2 = rec.empty(M, Cc)
                rec.rowchain(_lib.CHAIN_MID, Cc, M, HW, a, w, v, h1, out1=q2, res=h0)
                a = rec.empty(M, Cc)
                rec.attention(q2, ck, cvt, a, B, self.heads, d, HW, T, Cc, Cc, ldc_vt, Cc, HW * Cc, T * Cc, Cc * ldc_vt, HW * Cc, scale)
            h = h1
        out = rec.empty(M, Cc)
        part = rec.new_tot(B, Cc)                            # GroupNorm statistics of the block output (the next ResBlock's norm1)
        kw = {}
        r2kw = self._r2(r2, x.H, x.W)                      # (records the wait for the BlobNet branch's residual)
        if r2kw:
            assert r2kw["ldr2"] == Cc
            kw.update(r2=r2kw["R2"], r2_xmin=r2kw["r2_xmin"], r2_bmod=r2kw["r2_bmod"], out_w=r2kw["out_w"])
        res_out = None
        if zero is not None:
            zname, alpha, alpha_dev, alpha_idx, alpha_bstride = zero
            res_out = rec.empty(M, Cc)
            kw.update(out1=res_out, alpha=alpha, alpha_dev=alpha_dev, alpha_idx=alpha_idx, alpha_bstride=alpha_bstride)
        zname = zero[0] if zero is not None else None
        nsplit = self.rowchain_ff_split(Cc, M)
        if nsplit > 1 and self.rowchain_ffp():
            # the block end as OUT_FFP + sum (round 5): every slice goes on through proj_out [and the zero-conv] on its own partial sum
            # (both linear); what remains is an elementwise sum of nsplit fp16 partial outputs, with the GroupNorm statistics
            pp = rec.empty((2 if zero is not None else 1) * nsplit, M, Cc)
            w, v = packed(_lib.CHAIN_OUT_FFP, zname, nsplit)
            kw2 = {k: v_ for k, v_ in kw.items() if k != "out1"}
            rec.rowchain(_lib.CHAIN_OUT_FFP, Cc, M, HW, a, w, v, None, out1=res_out, res=h, res2=x.t, part=pp, nsplit=nsplit, **kw2)
            rec.rowchain_sum(Cc, M, HW, pp, nsplit, out, gn_tot=part, out1=res_out)
        elif nsplit > 1:
            # the block end as two launches, the feed-forward's hidden chunks spread over `nsplit` workgroups per row block
            ffp = rec.empty(nsplit, M, Cc, dtype=torch.float32)
            w, v = packed(_lib.CHAIN_OUT_FF, zname, nsplit)
            rec.rowchain(_lib.CHAIN_OUT_FF, Cc, M, HW, a, w, v, None, res=h, part=ffp, nsplit=nsplit)
            w, v = packed(_lib.CHAIN_OUT_TAIL, zname)
            rec.rowchain(_lib.CHAIN_OUT_TAIL, Cc, M, HW, None, w, v, out, res2=x.t, gn_tot=part, part=ffp, nsplit=nsplit, **kw)
        else:
            w, v = packed(_lib.CHAIN_OUT, zname)
            rec.rowchain(_lib.CHAIN_OUT, Cc, M, HW, a, w, v, out, res=h, res2=x.t, gn_tot=part, **kw)
        rec.tots[out.data_ptr()] = part
        return Act(out, Cc, x.H, x.W), res_out

    def gw_tile(self, M, N, K, C1=0, prefer=None):
        """BC_TILE_GW* configuration for a projection of the low-resolution levels (csrc/gemm_wreg.hip), or 0 = the LDS-DMA tiles.
        Measured per shape on an MI355X (tools/gw_probe.py, cold weights, graph replay): at M <= 1024 every projection is latency-bound
        (~11 us whatever the kernel), so the choice only matters where a launch disappears with it (LayerNorm folded, q | k | V^T in one
        launch) or the grid fills the chip (N >= 2560)."""
        if not opt("gw") or M > opt("gw_maxm"):
            return 0
        order = prefer or (_lib.TILE_GW64x128,)
        for cfg in order:
            if self.rec.lib.bc_gemm_wreg_eligible(M, N, K, C1, cfg):
                return cfg
        return 0

    def ctx_fold_ok(self, Cc, T):
        """The cross-attention of a block as two per-image projections (bc_ctx_fold): for the widths transformer_gw takes; 8 heads x 80 kept
        keys = K of the second projection (% 320).  BC_PLAN ctx_fold=0: to_q + bc_attention + to_out.
        Only for the single edit (UNet batch 2 = the CFG pair): every image brings its own 5.9 MB of folded weights per block, and from four
        images on the shared to_q / to_out weights win (same box, interleaved, ms per step, unfolded vs folded: one edit 9.155 / 9.098 vs 9.097 /
        9.076; two 15.715 / 15.583 vs 15.755 / 15.750; eight 52.75 / 52.58 vs 52.80 / 52.79; 768^2 x 4 79.85 / 79.61 vs 80.21 / 79.90)."""
        return (self.B <= opt("ctx_fold_maxb") and Cc % 320 == 0 and Cc not in (320, 640) and self.heads == 8 and T <= 80 and (Cc // self.heads) % 8 == 0 and Cc // self.heads <= 160
                and opt("ctx_fold") and opt("gw"))

    def transformer_gw(self, p, x: Act, r2=None):
        """One Transformer2D block of the 1280-channel levels on gemm_wreg.hip: the three LayerNorms are folded into the projections
        behind them (no LayerNorm launch, no normalised activation in HBM), q | k | V^T come out of ONE launch: 15 launches -> 11."""
        rec, pw, B = self.rec, self.pw, self.B
        Cc, HW = x.C, x.H * x.W
        M = B * HW
        d = Cc // self.heads
        scale = d ** -0.5
        bp = p + "transformer_blocks.0."
        G128, G256, G320 = _lib.TILE_GW64x128, _lib.TILE_GW64x256, _lib.TILE_GW64x320

        def proj(a_t, wname, N, K, cfg, ln=None, bias=True, extra=(), **kw):
            for c in (cfg, G128, G320, G256):                 # (the preferred workgroup shape, else one that divides N and the split point)
                bn = 64 * _lib.GW_TILES[c]
                if N % bn == 0 and kw.get("n_t0", 0) % bn == 0 and rec.lib.bc_gemm_wreg_eligible(M, N, K, kw.get("C1", 0), c):
                    cfg = c
                    break
            w, cs, b = pw.gw(wname + ".weight", cfg, ln=ln, bias=(wname + ".bias") if bias else None, extra=extra)
            n_out = N // 2 if kw.get("act") == _lib.ACT_GEGLU else N
            out = kw.pop("out", None)
            if out is None:
                out = rec.empty(M, kw.get("n_t0") or n_out)
            rec.gemm(A=a_t, W=w, M=M, N=N, K=K, out=out, bias=b, tile_cfg=cfg, ln_colsum=cs, **kw)
            return out
        # GroupNorm finalized + applied inside proj_in's row staging
        gn = dict(x1=x.t, C1=Cc, B=B, HW=HW, G=self.G, eps=1e-6, gamma=pw.f[p + "norm.weight"], beta=pw.f[p + "norm.bias"])
        h = proj(x.t, p + "proj_in", Cc, Cc, G128, kind="conv1x1", a_gn=gn, rows_per_batch=HW)
        # --- self attention: LayerNorm1 folded; q | k row-major, V transposed for the attention kernel
        ldvt = (HW + 63) // 64 * 64
        vt = rec.zeros(B, Cc, ldvt)
        qk = proj(h, bp + "attn1.to_qk", 3 * Cc, Cc, G128, ln=bp + "norm1", bias=False, extra=(bp + "attn1.to_v.weight",), C_t=vt, ldc_t=ldvt,
                  n_t0=2 * Cc, rows_per_batch=HW, kind="qkv")
        a = rec.empty(M, Cc)
        rec.attention(qk, qk, vt, a, B, self.heads, d, HW, HW, 2 * Cc, 2 * Cc, ldvt, Cc, HW * 2 * Cc, HW * 2 * Cc, Cc * ldvt, HW * Cc,
                      scale, q_off=0, k_off=Cc)
        h = proj(a, bp + "attn1.to_out.0", Cc, Cc, G128, R=h, ldr=Cc, kind="attn_out")
        folded = getattr(self, "ctx_folded", {}).get(bp) if pw.has_cross else None
        if folded is not None and rec.lib.bc_gemm_wreg_eligible(M, 128 * self.heads, Cc, 0, G128) and rec.lib.bc_gemm_wreg_eligible(M, Cc, 80 * self.heads, 0, G128):
            # cross-attention with the prompt folded into per-image weights, once per edit: softmax(LN2(h) QK^T) in one launch (a 64 x 128
            # workgroup = one head's keys, padded; the scores never leave it), then the 80 kept probabilities per head x VO + bias + residual
            wqk, cs, qb, vwo = folded
            T = self.ctx_kv[bp][2]
            N, Np = 128 * self.heads, 80 * self.heads
            pr = rec.empty(M, Np)
            rec.gemm(A=h, W=wqk, M=M, N=N, K=Cc, out=pr, ldc=Np, bias=qb, tile_cfg=G128, ln_colsum=cs, rows_per_batch=HW, w_bstride=wqk.shape[1],
                     vec_bstride=N, sm_group=128, sm_valid=T, sm_keep=80, kind="xattn")
            out = rec.empty(M, Cc)
            rec.gemm(A=pr, W=vwo, M=M, N=Cc, K=Np, out=out, bias=pw.f[bp + "attn2.to_out.0.bias"], tile_cfg=G128, R=h, ldr=Cc, rows_per_batch=HW,
                     w_bstride=vwo.shape[1], kind="xattn")
            h = out
        elif pw.has_cross:
            q = proj(h, bp + "attn2.to_q", Cc, Cc, G128, ln=bp + "norm2", bias=False, kind="qkv")
            ck, cvt, T, ldc_vt = self.ctx_kv[bp]
            a = rec.empty(M, Cc)
            rec.attention(q, ck, cvt, a, B, self.heads, d, HW, T, Cc, Cc, ldc_vt, Cc, HW * Cc, T * Cc, Cc * ldc_vt, HW * Cc, scale)
            h = proj(a, bp + "attn2.to_out.0", Cc, Cc, G128, R=h, ldr=Cc, kind="attn_out")
        # --- GEGLU feed-forward: LayerNorm3 folded into ff.net.0; ff.net.2 (K = 4C) stays on the LDS-DMA tiles with split-K
        # (64 x 128 workgroups everywhere: two fit a CU - 80 KiB of LDS each - and measure best at every shape of these levels, e.g.
        #  LayerNorm + GEGLU [1024 x 10240 x 1280] 51.7 us against 53.5 / 54.8 for the 256- / 320-column workgroups: tools/gw_probe.py)
        if opt("gw_ff1_g256") and self.g256_tile(M, 8 * Cc, Cc, dict(act=_lib.ACT_GEGLU)):
            # (round 6) ff.net.0 is the one projection of these levels wide enough for the 256 x 256 tiles (M = 1024: 4 x 40 = 160 workgroups of
            # one tile each): a LayerNorm launch + gemm256.hip with the GEGLU epilogue against the LayerNorm-folded gemm_wreg launch
            ln = rec.layernorm(h, M, Cc, pw.f[bp + "norm3.weight"], pw.f[bp + "norm3.bias"], 1e-5)
            g = self.dense(ln, M, Cc, bp + "ff.net.0.proj", 8 * Cc, act=_lib.ACT_GEGLU, kind="ff")
        else:
            g = proj(h, bp + "ff.net.0.proj", 8 * Cc, Cc, G128, ln=bp + "norm3", act=_lib.ACT_GEGLU, kind="ff")
        if rec.lib.bc_gemm_wreg_eligible(M, Cc, 5 * Cc, 4 * Cc, G128):
            # ff.net.2 + residual + proj_out as ONE two-source GEMM over [g | h] with the weight [P F2 | P] made at pack time
            # (weights.ff2_proj_out): at these levels a launch on the UNet's queue costs ~40 us INSIDE the step - three times what it
            # takes alone (tools/ablate_probe.py: the six attn2.to_q launches 0.245 ms) - and this one is pure algebra
            k = pw.ff2_proj_out(p)
            out = proj(g, k, Cc, 5 * Cc, G128, A2=h, C1=4 * Cc, lda=4 * Cc, lda2=Cc, R=x.t, ldr=Cc, kind="ff", rows_per_batch=HW, want_gn=True,
                       **self._r2(r2, x.H, x.W))
            return Act(out, Cc, x.H, x.W), None
        # ff.net.2 (K = 4C) unsplit on gemm_wreg: the same step time as the LDS-DMA tiles with split-K 3 (9.42 vs 9.42 ms, same box,
        # two rounds), without their 15.7 MB of fp32 slabs and the reducer launch
        h = proj(g, bp + "ff.net.2", Cc, 4 * Cc, G128, R=h, ldr=Cc, kind="ff")
        out = proj(h, p + "proj_out", Cc, Cc, G128, R=x.t, ldr=Cc, kind="conv1x1", rows_per_batch=HW, want_gn=True, **self._r2(r2, x.H, x.W))
        return Act(out, Cc, x.H, x.W), None

    def transformer(self, p, x: Act, r2=None, zero=None):
        """transformer_2d.py:479-527 + attention.py:421-541 (one BasicTransformerBlock).  Returns (output, zero-conv residual or None)."""
        rec, pw, B = self.rec, self.pw, self.B
        Cc, HW = x.C, x.H * x.W
        M = B * HW
        if self.rowchain_ok(Cc, M, HW, p):
            return self.transformer_rowchain(p, x, r2, zero)
        if Cc % 320 == 0 and HW % 64 == 0 and self.gw_tile(M, Cc, Cc):
            return self.transformer_gw(p, x, r2)
        d = Cc // self.heads
        scale = d ** -0.5
        bp = p + "transformer_blocks.0."
        n = self.groupnorm(x, None, p + "norm", 1e-6, False)
        h = self.dense(n.t, M, Cc, p + "proj_in", Cc, kind="conv1x1")
        # --- self attention
        ln = rec.layernorm(h, M, Cc, pw.f[bp + "norm1.weight"], pw.f[bp + "norm1.bias"], 1e-5)
        ldvt = (HW + 63) // 64 * 64
        vt = rec.zeros(B, Cc, ldvt)
        qkv_kw = dict(C_t=vt, ldc_t=ldvt, n_t0=2 * Cc, rows_per_batch=HW)
        if self.g256_tile(M, 3 * Cc, Cc, qkv_kw):
            # q | k row-major + V^T out of ONE launch on the 256 x 256 tiles (480 tiles at M = 8192: two full rounds of the chip instead of
            # 320 + 160 in three)
            qk = rec.empty(M, 2 * Cc)
            rec.gemm(A=ln, W=pw.h[pw.qkv_weight(bp)], M=M, N=3 * Cc, K=Cc, out=qk, tile_cfg=_lib.TILE_G256, kind="qkv", **qkv_kw)
        else:
            qk = self.dense(ln, M, Cc, None, 2 * Cc, bias=False, wkey=bp + "attn1.to_qk.weight", kind="qkv")
            self.dense(ln, M, Cc, None, Cc, bias=False, wkey=bp + "attn1.to_v.weight", out=vt, out_mode=_lib.OUT_F16_T,
                       ldc=ldvt, rows_per_batch=HW, kind="qkv")
        a = rec.empty(M, Cc)
        rec.attention(qk, qk, vt, a, B, self.heads, d, HW, HW, 2 * Cc, 2 * Cc, ldvt, Cc, HW * 2 * Cc, HW * 2 * Cc,
                      Cc * ldvt, HW * Cc, scale, q_off=0, k_off=Cc)
        h = self.dense(a, M, Cc, bp + "attn1.to_out.0", Cc, R=h, ldr=Cc, kind="attn_out")
        # --- cross attention (UNet only)
        if pw.has_cross:
            ln = rec.layernorm(h, M, Cc, pw.f[bp + "norm2.weight"], pw.f[bp + "norm2.bias"], 1e-5)
            q = self.dense(ln, M, Cc, None, Cc, bias=False, wkey=bp + "attn2.to_q.weight", kind="qkv")
            ck, cvt, T, ldc_vt = self.ctx_kv[bp]
            a = rec.empty(M, Cc)
            rec.attention(q, ck, cvt, a, B, self.heads, d, HW, T, Cc, Cc, ldc_vt, Cc, HW * Cc, T * Cc, Cc * ldc_vt,
                          HW * Cc, scale)
            h = self.dense(a, M, Cc, bp + "attn2.to_out.0", Cc, R=h, ldr=Cc, kind="attn_out")
        # --- GEGLU feed-forward
        ln = rec.layernorm(h, M, Cc, pw.f[bp + "norm3.weight"], pw.f[bp + "norm3.bias"], 1e-5)
        g = self.dense(ln, M, Cc, bp + "ff.net.0.proj", 8 * Cc, act=_lib.ACT_GEGLU, kind="ff")
        r2kw = self._r2(r2, x.H, x.W)
        fused_kw = dict(A2=h, C1=4 * Cc, lda=4 * Cc, lda2=Cc, R=x.t, ldr=Cc, rows_per_batch=HW, want_gn=True, **r2kw)
        if self.g256_tile(M, Cc, 5 * Cc, fused_kw):
            # ff.net.2 + residual + proj_out as ONE two-source GEMM over [g | h] with the pack-time product [P F2 | P] (weights.ff2_proj_out,
            # as on gemm_wreg.hip at M <= 1024): a dependent launch and the h round trip less
            k = pw.ff2_proj_out(p)
            out = rec.empty(M, Cc)
            rec.gemm(A=g, W=pw.h[k + ".weight"], M=M, N=Cc, K=5 * Cc, out=out, bias=pw.f[k + ".bias"], tile_cfg=_lib.TILE_G256, kind="ff", **fused_kw)
            return Act(out, Cc, x.H, x.W), None
        h = self.dense(g, M, 4 * Cc, bp + "ff.net.2", Cc, R=h, ldr=Cc, kind="ff")
        # --- proj_out + residual (+ BlobNet residual)
        out = self.dense(h, M, Cc, p + "proj_out", Cc, R=x.t, ldr=Cc, kind="conv1x1", rows_per_batch=HW, want_gn=True, **r2kw)
        return Act(out, Cc, x.H, x.W), None

    # ------------------------------------------------------------------------------------------- prologue
    def record_context(self, ctx: torch.Tensor, T: int):
        """Cross-attention K / V^T of the (step-invariant) prompt embeddings, once per edit
        (attention.py:504-510; SURVEY Appendix A 'step-invariant => precompute once per edit')."""
        rec, pw, B = self.rec, self.pw, self.B
        self.ctx_kv, self.ctx_kvs, self.ctx_folded = {}, {}, {}
        Dc = ctx.shape[-1]
        ldvt = (T + 63) // 64 * 64
        for k in [k for k in pw.h if k.endswith("attn2.to_k.weight")]:
            bp = k[: -len("attn2.to_k.weight")]
            Cc = pw.h[k].shape[0]
            ck = rec.empty(B * T, Cc)
            rec.gemm(A=ctx, W=pw.h[k], M=B * T, N=Cc, K=Dc, out=ck, kind="ctx_kv")
            cvt = rec.zeros(B, Cc, ldvt)
            rec.gemm(A=ctx, W=pw.h[bp + "attn2.to_v.weight"], M=B * T, N=Cc, K=Dc, out=cvt, out_mode=_lib.OUT_F16_T,
                     ldc=ldvt, rows_per_batch=T, kind="ctx_kv")
            self.ctx_kv[bp] = (ck, cvt, T, ldvt)
            # blocks the row-chain takes with 8 heads and <= 80 context tokens: K / V^T also as the fragment streams of CHAIN_MIDX
            if Cc in (320, 640) and self.heads == 8 and T <= 80 and opt("midx") and opt("rowchain"):
                self.ctx_kvs[bp] = rec.rowchain_kv_stream(ck, cvt, B, T, Cc, ldvt)
            # blocks on gemm_wreg.hip (the 1280-channel levels): the prompt folded into attn2's weights (bc_ctx_fold) - to_q + attention +
            # to_out become two projections with per-image weights
            if self.ctx_fold_ok(Cc, T):
                wq, bq, wo = pw.ctx_fold_weights(bp)
                self.ctx_folded[bp] = rec.ctx_fold(ck, cvt, B, T, Cc, ldvt, self.heads, (Cc // self.heads) ** -0.5, wq, bq, wo)

    # ------------------------------------------------------------------------------------------- per-edit weight collapse
    def record_collapse(self, feat16: torch.Tensor, per_image: int = 0):
        """BlobNet conv_in over the F feature channels, which are score x f (rank 1, pipe:706-721, SURVEY 8a note iii):
        g[o, tap] = sum_c W[o, 5+c, tap] * f_c is computed once per edit (one GEMV on the MFMA path) and written into input
        channel 5 of an 8-channel conv_in weight; the step then convolves [latents(4), score, score, 0, 0] - as a dense K = 128
        GEMM over the im2col operand that bc_assemble_input_im2col writes (weight [co][128], k = tap * 8 + channel).
        `per_image` = B (round 6): a batch of B independent requests has B feature vectors, hence B collapsed weights - one GEMV per
        image into its own copy of the 8-channel weight, and conv_in then runs as B launches of one image each (record_forward) instead
        of ONE 1029-channel convolution over the batch: 390 GFLOP per step (K = 9288, off the LDS-DMA fast path) become 8 x 0.7 GFLOP."""
        pw = self.pw
        w8 = pw.h["conv_in.weight8"]
        fm = pw.h["conv_in.featmat"]
        if per_image:
            # (a plan-owned copy per image, saved WITH its contents by bc_plan_save: channels 0-4 of the weight are constants)
            self.w8_img = self.rec.register(w8.unsqueeze(0).repeat(per_image, 1, 1).contiguous())
            self.rec.keep.append(self.w8_img)
            for b in range(per_image):
                self.rec.gemm(A=fm, W=feat16[b:b + 1], M=fm.shape[0], N=1, K=fm.shape[1], out=self.w8_img[b], ldc=8, out_offset=5, kind="collapse")
            return
        self.rec.gemm(A=fm, W=feat16, M=fm.shape[0], N=1, K=fm.shape[1], out=w8, ldc=8, out_offset=5, kind="collapse")

    # ------------------------------------------------------------------------------------------- time embedding
    def record_time(self, t_table, t_idx, t_value=0.0):
        """embeddings.py:27-78, 576-588 and the 22 `time_emb_proj(silu(emb))` of resnet.py:343-350 as three GEMMs."""
        rec, pw, B = self.rec, self.pw, self.B
        c0 = self.cfg.block_out_channels[0]
        te = c0 * 4
        sin = rec.empty(B, c0)
        rec.call("bc_timestep_embedding", _ptr(t_table), _ptr(t_idx), float(t_value), B, c0, sin.data_ptr(),
                 kind="temb", keep=(t_table, t_idx, sin))
        h = self.dense(sin, B, c0, "time_embedding.linear_1", te, act=_lib.ACT_SILU, kind="temb")
        h = self.dense(h, B, te, "time_embedding.linear_2", te, act=_lib.ACT_SILU, kind="temb")   # silu(emb)
        self.tproj = self.dense(h, B, te, "temb_all", pw.temb_total, kind="temb")

    def record_time_table(self, t_table, nsteps, t_idx):
        """The same three GEMMs over ALL steps of an edit at once (prologue): tproj [nsteps * B][temb_total]; the ResBlock
        epilogues then pick the block of the current step through the device step counter (BcGemm.rowvec_idx), and no
        time-embedding launch is left in the step."""
        rec, pw, B = self.rec, self.pw, self.B
        c0 = self.cfg.block_out_channels[0]
        te = c0 * 4
        rows = nsteps * B
        sin = rec.empty(rows, c0)
        rec.call("bc_timestep_embedding_table", _ptr(t_table), nsteps, B, c0, sin.data_ptr(), kind="temb", keep=(t_table, sin))
        h = self.dense(sin, rows, c0, "time_embedding.linear_1", te, act=_lib.ACT_SILU, kind="temb")
        h = self.dense(h, rows, te, "time_embedding.linear_2", te, act=_lib.ACT_SILU, kind="temb")
        self.tproj = self.dense(h, rows, te, "temb_all", pw.temb_total, kind="temb")
        self.tproj_table = (t_idx, B * pw.temb_total)

    # ------------------------------------------------------------------------------------------- forward
    def conv_in_dense(self, x_in: torch.Tensor, wkey, r2=None, out=None, gn_tot_out=None):
        """conv_in on the im2col operand [B][H*W][128] (round 3): K = 72 is outside the LDS-DMA GEMM's fast path (the register-staged
        kernel took 41 us per launch for 2.4 GFLOP); as a dense K = 128 GEMM it is one of the ordinary 1x1-convolution launches."""
        pw, H, W = self.pw, self.H, self.W
        Cout = self.cfg.block_out_channels[0]
        M = self.B * H * W
        if out is None:
            out = self.rec.empty(self.B, H * W, Cout)
        self.rec.gemm(A=x_in, W=pw.h[wkey], M=M, N=Cout, K=128, out=out, bias=pw.f["conv_in.bias"], rows_per_batch=H * W, kind="conv_in",
                      want_gn=True, gn_tot_out=gn_tot_out, **self._r2(r2, H, W))
        return Act(out, Cout, H, W)

    def cfg_prefix_ok(self, cfg_pairs, im2col):
        """The CFG-invariant prefix of the UNet runs once per image PAIR (round 6).  Both classifier-free-guidance images enter the UNet
        identical (pipe:1031 `torch.cat([latents] * 2)`, the same BlobNet residuals, the same timestep: pipe:1071-1076) and the prompt first
        enters at attn2 of down_blocks.0.attentions.0 (attention.py:504-510), so conv_in, the residual add, down_blocks.0.resnets.0 and
        that block's GroupNorm -> proj_in -> norm1 -> attn1 are the same arithmetic on the same numbers for images b and b + B: 125 of the
        step's 5578 GFLOP at 512^2, half of one 8192-token self-attention among them.  They are recorded at batch B; one bc_dup_halves
        launch fans the four tensors that later launches read per image out to the pairs.  Results are bit-identical to the plain plan
        - to fp16 rounding, tests/test_fullsize_gpu.py.  BC_PLAN cfg_prefix=0: the plain plan."""
        cfg = self.cfg
        if not cfg_pairs or cfg.is_blobnet or self.B % 2 or not im2col or not opt("cfg_prefix") or len(cfg.block_out_channels) < 2:
            return False
        Cc, HW = cfg.block_out_channels[0], self.H * self.W
        return self.pw.has_cross and Cc in (320, 640) and self.rowchain_ok(Cc, (self.B // 2) * HW, HW, "down_blocks.0.attentions.0.")

    def record_forward(self, x_in: torch.Tensor, residuals: Optional[Residuals] = None, zero_scale=None,
                       signal_residuals: bool = False, eps_out: Optional[torch.Tensor] = None, im2col: bool = False, cfg_pairs: bool = False):
        """x_in: [B, H*W, pad8(in_channels)] fp16, or with `im2col` the [B, H*W, 128] operand of bc_assemble_input_im2col.  UNet: returns eps fp32 [B, H*W, out_channels].
        BlobNet: returns Residuals (zero-conv outputs times `zero_scale` = (alpha, alpha_dev, alpha_idx[, alpha_bstride]);
        alpha_bstride = B selects per-image scales alpha_dev[step * B + image] for a batch of independent requests)."""
        cfg, pw = self.cfg, self.pw
        boc = cfg.block_out_channels
        nb = len(boc)
        H, W = self.H, self.W
        res_d = list(residuals.down) if residuals is not None else None
        res_u = list(residuals.up) if residuals is not None else None
        self.res_bmod = residuals.bmod if residuals is not None else 1
        self.res_events = residuals.events if residuals is not None else None
        pop = (lambda lst: lst.pop(0)) if residuals is not None else (lambda lst: None)
        alpha, alpha_dev, alpha_idx, alpha_bstride = (tuple(zero_scale) + (0,))[:4] if zero_scale is not None else (1.0, None, None, 0)
        out_events = {} if signal_residuals else None

        class _Feats(list):
            """BlobNet: every trunk feature is turned into its residual (zero-conv 1x1, bn:860-864, 881, 921-924, times
            conditioning_scale, bn:936-938) as soon as it exists, so the UNet branch can consume it while BlobNet runs on."""
            def __init__(s2, prefix):
                super().__init__()
                s2.prefix, s2.res = prefix, []

            def next_zero(s2):
                """(name, scale arguments) of the zero-conv that the NEXT appended feature goes through (BlobNet), else None."""
                if not cfg.is_blobnet:
                    return None
                name = s2.prefix if s2.prefix.endswith("mid_block") else f"{s2.prefix}.{len(s2)}"
                return (name, alpha, alpha_dev, alpha_idx, alpha_bstride)

            def append(s2, f, pre=None):
                """`pre`: the residual already produced by the fused row-chain (zero-conv inside the block's last launch)."""
                super().append(f)
                if cfg.is_blobnet:
                    name = s2.prefix if s2.prefix.endswith("mid_block") else f"{s2.prefix}.{len(s2) - 1}"
                    M = self.B * f.H * f.W
                    # (the low-resolution zero-convs on gemm_wreg.hip measured the same step time: 9.43 vs 9.43 ms; not taken)
                    r = pre if pre is not None else \
                        self.dense(f.t, M, f.C, name, f.C, kind="zero_conv", alpha=alpha, alpha_dev=alpha_dev,
                                   alpha_idx=alpha_idx, alpha_bstride=alpha_bstride, rows_per_batch=f.H * f.W)
                    r = r.view(self.B, f.H * f.W, f.C)
                    if out_events is not None:
                        ev = self.rec.new_event()
                        self.rec.signal(ev)
                        out_events[r.data_ptr()] = ev
                    s2.res.append(r)

        feats_d, feats_u, feats_m = _Feats("blobnet_down_blocks"), _Feats("blobnet_up_blocks"), _Feats("blobnet_mid_block")

        x = Act(x_in, x_in.shape[-1], H, W)
        conv_in_name = "conv_in"
        if not im2col and x_in.shape[-1] == 8 and pad8(cfg.in_channels) > 8:
            raise ValueError("the rank-1-collapsed BlobNet input is consumed in its im2col form (bc_assemble_input_im2col, im2col=True)")
        if im2col and getattr(self, "w8_img", None) is not None:
            # per-request batch (record_collapse per_image): one launch per image with that image's collapsed weight; the statistics totals
            # of the output are one table over the batch, every launch adding to its own image's rows
            def conv_in(r2=None):
                assert r2 is None                              # (BlobNet takes no residuals)
                out_all, tot = self.rec.empty(self.B, H * W, boc[0]), self.rec.new_tot(self.B, boc[0])
                for b in range(self.B):
                    self.rec.gemm(A=x_in[b], W=self.w8_img[b], M=H * W, N=boc[0], K=128, out=out_all[b], bias=pw.f["conv_in.bias"],
                                  rows_per_batch=H * W, kind="conv_in", want_gn=True, gn_tot_out=tot[b:b + 1])
                self.rec.tots[out_all.data_ptr()] = tot
                return Act(out_all, boc[0], H, W)
        elif im2col:
            wkey = "conv_in.weight8" if pad8(cfg.in_channels) > 8 else "conv_in.weight_k128"
            conv_in = lambda r2=None: self.conv_in_dense(x_in, wkey, r2)
        else:
            conv_in = lambda r2=None: self.conv3x3(x, conv_in_name, boc[0], r2=r2, kind="conv_in")
        prefix = self.cfg_prefix_ok(cfg_pairs, im2col)       # `cfg_pairs`: images b and b + B / 2 of x_in are identical (the CFG pair)
        if prefix:
            # ---- the CFG-invariant prefix at HALF the batch (cfg_prefix_ok): conv_in, residual add, down_blocks.0.resnets.0 and the head of
            # down_blocks.0.attentions.0 up to its self-attention; full-size buffers whose first half the half-batch launches write
            Bf, Bh, rec = self.B, self.B // 2, self.rec
            skip_full, hres_full = rec.empty(Bf, H * W, boc[0]), rec.empty(Bf, H * W, boc[0])
            skip_tot = rec.new_tot(Bf, boc[0])               # skip #0's GroupNorm statistics (read by up_blocks.3.resnets.2 per image)
            self.B = Bh
            try:
                xh = x_in[:Bh]
                if residuals is not None and W == H:           # (the two canvas cases as below)
                    skip0 = self.conv_in_dense(xh, wkey, None, out=skip_full[:Bh], gn_tot_out=skip_tot[:Bh])
                    h = self.conv_in_dense(xh, wkey, pop(res_d))
                else:
                    h = self.conv_in_dense(xh, wkey, pop(res_d), out=skip_full[:Bh], gn_tot_out=skip_tot[:Bh])
                r = pop(res_d)
                h = self.resnet("down_blocks.0.resnets.0.", h, None, boc[0], r2=None, out=hres_full[:Bh])
                half = Bh * H * W * boc[0] * 2
                h, _ = self.transformer_rowchain("down_blocks.0.attentions.0.", h, r2=r, zero=None,
                                                 fan_out=(hres_full, [(skip_full, half), (skip_tot, skip_tot[:Bh].numel() * 8)]))
            finally:
                self.B = Bf
            rec.tots[skip_full.data_ptr()] = skip_tot
            skips = [Act(skip_full, boc[0], H, W), h]
        else:
            if residuals is not None and W == H:
                # square canvas: `sample = sample + r` rebinds, skip #0 stays WITHOUT the residual (unet_2d_condition.py:1213-1217)
                skip0 = conv_in()
                h = conv_in(pop(res_d))
            else:
                # wide canvas: in-place slice add aliases the tuple element => skip #0 carries the residual (:1219)
                h = conv_in(pop(res_d))
                skip0 = h
            skips = [skip0]
            feats_d.append(h)
        for i in range(nb):
            has_attn = i < nb - 1
            for j in range(cfg.layers_per_block):
                if prefix and i == 0 and j == 0:
                    continue                                   # (recorded above)
                r = pop(res_d)
                h = self.resnet(f"down_blocks.{i}.resnets.{j}.", h, None, boc[i], r2=None if has_attn else r)
                pre = None
                if has_attn:
                    h, pre = self.transformer(f"down_blocks.{i}.attentions.{j}.", h, r2=r, zero=feats_d.next_zero())
                skips.append(h)
                feats_d.append(h, pre)
            if i < nb - 1:
                h = self.conv3x3(h, f"down_blocks.{i}.downsamplers.0.conv", boc[i], stride=2, r2=pop(res_d),
                                 kind="downsample")
                skips.append(h)
                feats_d.append(h)
        # mid (unet_2d_blocks.py:860-899) ; residual after the block (unet_2d_condition.py:1292-1296)
        h = self.resnet("mid_block.resnets.0.", h, None, boc[-1])
        h, _ = self.transformer("mid_block.attentions.0.", h)
        h = self.resnet("mid_block.resnets.1.", h, None, boc[-1], r2=residuals.mid if residuals is not None else None)
        feat_mid = h
        feats_m.append(h)
        rev = list(reversed(boc))
        for i in range(nb):
            has_attn = i > 0
            n_res = cfg.layers_per_block + 1
            res = skips[-n_res:]
            del skips[-n_res:]
            for j in range(n_res):
                r = pop(res_u)
                sk = res.pop()
                h = self.resnet(f"up_blocks.{i}.resnets.{j}.", h, sk, rev[i], r2=None if has_attn else r)
                pre = None
                if has_attn:
                    h, pre = self.transformer(f"up_blocks.{i}.attentions.{j}.", h, r2=r, zero=feats_u.next_zero())
                feats_u.append(h, pre)
            if i < nb - 1:
                size = (skips[-1].H, skips[-1].W)
                h = self.conv3x3(h, f"up_blocks.{i}.upsamplers.0.conv", rev[i], up_to=size, r2=pop(res_u),
                                 kind="upsample")
                feats_u.append(h)
        if not cfg.is_blobnet:
            n = self.groupnorm(h, None, "conv_norm_out", 1e-5, True)
            # conv_out has 4 output channels: one column tile.  The planner's 256-row tiles left 64 workgroups, each gathering 1.5 MB
            # through one CU (23 us); 64-row tiles spread the same gather over 256 CUs.
            small = 7 if (self.B * H * W) >= 64 * 128 else 0
            eps = self.conv3x3(n, "conv_out", cfg.out_channels, out_f32=True, kind="conv_out", out=eps_out, tile_cfg=small)
            return eps.t
        self.feat_shapes = ([(f.C, f.H, f.W) for f in feats_d], (feat_mid.C, feat_mid.H, feat_mid.W),
                            [(f.C, f.H, f.W) for f in feats_u])
        return Residuals(feats_d.res, feats_m.res[0], feats_u.res, bmod=self.B, events=out_events)


def _ptr(t):
    return None if t is None else t.data_ptr()
