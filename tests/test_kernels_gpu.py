"""Kernel-level parity on MI355X: every C-ABI entry point against a plain PyTorch fp32 statement of the same op
(computed on CPU from the SAME fp16-rounded inputs).  Tolerances: fp16 output rounding (rel 1e-3) + fp32 accumulation
order; attention 2e-3 abs on O(1) outputs."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.common import g  # noqa: E402


@pytest.fixture(scope="module")
def rec():
    from blobctrl_amd.launch import Recorder
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    r = Recorder(torch.device("cuda:0"))
    return r


def run(rec, fn):
    seg = rec.begin("test")
    out = fn()
    seg.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def h(x):
    return x.half().cuda()


def close(a, b, rtol=2e-3, atol=None, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    if atol is None:
        atol = 2e-3 * max(1.0, float(b.abs().max()))
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4e} (atol {atol:.2e})"


# ---------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 320, 320), (154, 320, 768), (2, 1280, 320), (4096, 640, 640),
                                    (33, 8, 8), (130, 72, 200)])
def test_gemm_dense_bias(rec, M, N, K):
    from blobctrl_amd import _lib
    A, W, b = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N)
    out = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda()))
    ref = A.half().float() @ W.half().float().t() + b
    close(out, ref, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("N,K,splitk", [(192, 128, None), (192, 1280, 3), (20, 72, None)])
def test_gemm_per_image_alpha(rec, N, K, splitk):
    """alpha_bstride: one device scalar per image and step (request batches with their own conditioning scales)."""
    from blobctrl_amd import _lib
    B, rows = 3, 100
    M = B * rows
    A, W, b = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N)
    tab = torch.tensor([9.0, 9.0, 9.0, 0.0, 0.7, 1.3]).cuda()          # [step][image]; step 1 is used
    idx = torch.tensor([1], dtype=torch.int32).cuda()
    out = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), alpha=2.0, alpha_dev=tab,
                                    alpha_idx=idx, alpha_bstride=B, rows_per_batch=rows, splitk=splitk))
    v = (A.half().float() @ W.half().float().t() + b) * 2.0
    ref = v * torch.tensor([0.0, 0.7, 1.3]).repeat_interleave(rows)[:, None]
    close(out, ref, what="per-image alpha")
    with pytest.raises(_lib.BlobCtrlHipError):                          # needs rows_per_batch
        run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), alpha_dev=tab, alpha_idx=idx, alpha_bstride=B))


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_gemm_act_residual_colscale_alpha(rec, act):
    from blobctrl_amd import _lib
    M, N, K = 300, 192, 128
    A, W, b, R, cs = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N), g(4, M, N), g(5, N)
    tab = torch.tensor([0.0, 0.7, 1.3]).cuda()
    idx = torch.tensor([1], dtype=torch.int32).cuda()
    out = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(),
                                    act=_lib.ACT_GELU if act == "gelu" else _lib.ACT_SILU, colscale=cs.cuda(), alpha=0.5,
                                    alpha_dev=tab, alpha_idx=idx, R=h(R), ldr=N))
    v = A.half().float() @ W.half().float().t() + b
    v = F.gelu(v) if act == "gelu" else F.silu(v)
    ref = v * cs * 0.5 * 0.7 + R.half().float()
    close(out, ref, what="epilogue chain")


@pytest.mark.parametrize("M,N,K,sk", [(256, 1280, 11520, 6), (128, 320, 1280, 4), (70, 64, 640, 3)])
def test_gemm_splitk(rec, M, N, K, sk):
    A, W, b, R = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N), g(4, M, N)
    out = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), R=h(R), ldr=N,
                                    splitk=sk))
    ref = A.half().float() @ W.half().float().t() + b + R.half().float()
    close(out, ref, what="splitk")


@pytest.mark.parametrize("C1,C2,N", [(8, 16, 24), (320, 640, 320), (72, 56, 100)])
def test_gemm_concat_sources(rec, C1, C2, N):
    M = 260
    A1, A2, W = g(1, M, C1), g(2, M, C2), g(3, N, C1 + C2) / math.sqrt(C1 + C2)
    out = run(rec, lambda: rec.gemm(A=h(A1), A2=h(A2), C1=C1, lda=C1, lda2=C2, W=h(W), M=M, N=N, K=C1 + C2,
                                    out=rec.empty(M, N)))
    ref = torch.cat([A1, A2], 1).half().float() @ W.half().float().t()
    close(out, ref, what="concat")


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("mode", ["dense", "dense_sk3", "concat", "conv1", "conv2", "ups", "transposed"])
def test_every_fast_tile_configuration(rec, cfg, mode):
    """Each compiled (tile shape, stage count) x operand mode of the fast kernel on ragged M / N (300 x 200), a K loop shorter than
    the deepest LDS ring (3 k-steps) and a 3-way split-K - the planner / tuner may pick any of them for a production shape."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_conv3x3
    if mode in ("dense", "dense_sk3", "concat", "transposed"):
        M, N, K = 300, 200, 640 if mode == "dense_sk3" else 192
        A, W, b = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N)
        ref = A.half().float() @ W.half().float().t() + b
        if mode == "concat":
            out = run(rec, lambda: rec.gemm(A=h(A[:, :128].contiguous()), A2=h(A[:, 128:].contiguous()), C1=128, lda=128, lda2=64,
                                            W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), tile_cfg=cfg, splitk=1))
        elif mode == "transposed":
            out = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.zeros(3, N, 104), bias=b.cuda(),
                                            out_mode=_lib.OUT_F16_T, ldc=104, rows_per_batch=100, tile_cfg=cfg, splitk=1))
            out, ref = out[:, :, :100], ref.view(3, 100, N).transpose(1, 2)
        else:
            out = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), tile_cfg=cfg,
                                            splitk=3 if mode == "dense_sk3" else 1))
        close(out, ref, what=f"{mode} cfg {cfg}")
        return
    B, Cc, Co, H, Wd = 2, 64, 72, 9, 11
    x, w, b = g(1, B, Cc, H, Wd), g(2, Co, Cc, 3, 3) / math.sqrt(9 * Cc), g(3, Co)
    if mode == "ups":
        Hv, Wv, stride = 2 * H, 2 * Wd, 1
        src = F.interpolate(x.half().float(), size=(Hv, Wv), mode="nearest")
    else:
        Hv, Wv, stride = H, Wd, 2 if mode == "conv2" else 1
        src = x.half().float()
    Ho, Wo = (Hv - 1) // stride + 1, (Wv - 1) // stride + 1
    out = run(rec, lambda: rec.gemm(A=nhwc(x), W=h(pack_conv3x3(w)), M=B * Ho * Wo, N=Co, K=9 * Cc, out=rec.empty(B, Ho * Wo, Co),
                                    bias=b.cuda(), conv=dict(Cin=Cc, Hin=H, Win=Wd, Hv=Hv, Wv=Wv, Hout=Ho, Wout=Wo, stride=stride),
                                    tile_cfg=cfg, splitk=1))
    close(from_nhwc(out, B, Ho, Wo), F.conv2d(src, w.half().float(), b, stride=stride, padding=1), what=f"{mode} cfg {cfg}")


def test_gemm_transposed_and_f32_out(rec):
    from blobctrl_amd import _lib
    B, T, N, K = 2, 77, 80, 64
    A, W = g(1, B * T, K), g(2, N, K) / math.sqrt(K)
    ldvt = 128
    vt = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=B * T, N=N, K=K, out=rec.zeros(B, N, ldvt), out_mode=_lib.OUT_F16_T,
                                   ldc=ldvt, rows_per_batch=T))
    ref = (A.half().float() @ W.half().float().t()).view(B, T, N).transpose(1, 2)
    close(vt[:, :, :T], ref, what="transposed")
    assert float(vt[:, :, T:].abs().max()) == 0.0
    o32 = run(rec, lambda: rec.gemm(A=h(A), W=h(W), M=B * T, N=N, K=K, out=rec.empty(B * T, N, dtype=torch.float32),
                                    out_mode=_lib.OUT_F32))
    assert o32.dtype == torch.float32
    close(o32, A.half().float() @ W.half().float().t(), rtol=1e-4, atol=1e-4, what="f32 out")


@pytest.mark.parametrize("C", [8, 40, 320])
def test_gemm_geglu(rec, C):
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import interleave_geglu
    M = 200
    A, W, b = g(1, M, C), g(2, 8 * C, C) / math.sqrt(C), g(3, 8 * C)
    wi, bi = interleave_geglu(W, b)
    out = run(rec, lambda: rec.gemm(A=h(A), W=h(wi), M=M, N=8 * C, K=C, out=rec.empty(M, 4 * C), bias=bi.cuda(),
                                    act=_lib.ACT_GEGLU))
    hg = A.half().float() @ W.half().float().t() + b
    val, gate = hg.chunk(2, -1)
    close(out, val * F.gelu(gate), what="geglu")
    out2 = run(rec, lambda: rec.gemm(A=h(A), W=h(wi), M=M, N=8 * C, K=C, out=rec.empty(M, 4 * C), bias=bi.cuda(),
                                     act=_lib.ACT_GEGLU, splitk=2)) if C >= 128 else out
    close(out2, val * F.gelu(gate), what="geglu splitk")


# ---------------------------------------------------------------------------------------------------- conv
def nhwc(x):      # NCHW fp32 -> token-major fp16 on device
    B, Cc, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, H * W, Cc).half().cuda().contiguous()


def from_nhwc(t, B, H, W):
    return t.float().cpu().view(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,Cin,Cout,H,W,stride", [(2, 8, 16, 8, 16, 1), (1, 64, 320, 16, 32, 1), (2, 320, 320, 16, 32, 2),
                                                   (1, 24, 40, 7, 9, 1), (1, 24, 40, 7, 9, 2), (2, 128, 4, 8, 16, 1)])
def test_conv3x3(rec, B, Cin, Cout, H, W, stride):
    from blobctrl_amd.weights import pack_conv3x3
    x, w, b = g(1, B, Cin, H, W), g(2, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(3, Cout)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = run(rec, lambda: rec.gemm(A=nhwc(x), W=h(pack_conv3x3(w)), M=B * Ho * Wo, N=Cout, K=9 * Cin,
                                    out=rec.empty(B, Ho * Wo, Cout), bias=b.cuda(),
                                    conv=dict(Cin=Cin, Hin=H, Win=W, Hout=Ho, Wout=Wo, stride=stride)))
    ref = F.conv2d(x.half().float(), w.half().float(), b, stride=stride, padding=1)
    close(from_nhwc(out, B, Ho, Wo), ref, what="conv3x3")


def test_conv3x3_padded_input_channels(rec):
    """conv_in: 5 real channels padded to 8 (UNet) - padded lanes carry zeros on both operands."""
    from blobctrl_amd.weights import pack_conv3x3
    B, Cin, Cout, H, W = 2, 5, 32, 8, 16
    x, w, b = g(1, B, Cin, H, W), g(2, Cout, Cin, 3, 3) / math.sqrt(45), g(3, Cout)
    xp = torch.zeros(B, 8, H, W)
    xp[:, :Cin] = x
    out = run(rec, lambda: rec.gemm(A=nhwc(xp), W=h(pack_conv3x3(w)), M=B * H * W, N=Cout, K=72,
                                    out=rec.empty(B, H * W, Cout), bias=b.cuda(),
                                    conv=dict(Cin=8, Hin=H, Win=W, Hout=H, Wout=W)))
    close(from_nhwc(out, B, H, W), F.conv2d(x.half().float(), w.half().float(), b, padding=1), what="conv_in")


@pytest.mark.parametrize("size", [None, (9, 13)])
def test_conv3x3_fused_upsample(rec, size):
    from blobctrl_amd.weights import pack_conv3x3
    B, Cc, H, W = 2, 64, 4, 8
    x, w, b = g(1, B, Cc, H, W), g(2, Cc, Cc, 3, 3) / math.sqrt(9 * Cc), g(3, Cc)
    Hv, Wv = size if size else (2 * H, 2 * W)
    out = run(rec, lambda: rec.gemm(A=nhwc(x), W=h(pack_conv3x3(w)), M=B * Hv * Wv, N=Cc, K=9 * Cc,
                                    out=rec.empty(B, Hv * Wv, Cc), bias=b.cuda(),
                                    conv=dict(Cin=Cc, Hin=H, Win=W, Hv=Hv, Wv=Wv, Hout=Hv, Wout=Wv)))
    up = F.interpolate(x.half().float(), size=(Hv, Wv), mode="nearest")
    close(from_nhwc(out, B, Hv, Wv), F.conv2d(up, w.half().float(), b, padding=1), what="upsample conv")


@pytest.mark.parametrize("H,W", [(8, 16), (8, 8)])
def test_conv_rowvec_residual_and_blobnet_right_half(rec, H, W):
    """conv1 (+time embedding row vector) and conv2 (+shortcut residual, + BlobNet residual on the right square, broadcast
    over the CFG halves: unet_2d_blocks.py:1303-1307 with `b % Bblob`)."""
    from blobctrl_amd.weights import pack_conv3x3
    B, Bb, Cc = 4, 2, 64
    x, w, b = g(1, B, Cc, H, W), g(2, Cc, Cc, 3, 3) / math.sqrt(9 * Cc), g(3, Cc)
    temb, R, R2 = g(4, B, 3 * Cc), g(5, B, Cc, H, W), g(6, Bb, Cc, H, W)
    xmin = 0 if H == W else W - H
    dtemb = h(temb)
    out = run(rec, lambda: rec.gemm(A=nhwc(x), W=h(pack_conv3x3(w)), M=B * H * W, N=Cc, K=9 * Cc,
                                    out=rec.empty(B, H * W, Cc), bias=b.cuda(),
                                    conv=dict(Cin=Cc, Hin=H, Win=W, Hout=H, Wout=W),
                                    rowvec=dtemb.data_ptr() + Cc * 2, ld_rowvec=3 * Cc, R=nhwc(R), ldr=Cc,
                                    R2=nhwc(R2), ldr2=Cc, r2_xmin=xmin, r2_bmod=Bb, out_w=W))
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1) + temb.half().float()[:, Cc:2 * Cc, None, None]
    ref = ref + R.half().float()
    add = R2.half().float().repeat(2, 1, 1, 1)
    ref[..., xmin:] = ref[..., xmin:] + add[..., xmin:]
    close(from_nhwc(out, B, H, W), ref, what="conv epilogue")


# ---------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,C1,C2,HW,G,eps,silu", [(2, 320, 0, 512, 32, 1e-5, True), (2, 1280, 640, 128, 32, 1e-5, True),
                                                   (1, 16, 0, 100, 4, 1e-6, False), (2, 32, 16, 37, 4, 1e-5, True),
                                                   (1, 640, 320, 2048, 32, 1e-5, True)])
def test_groupnorm(rec, B, C1, C2, HW, G, eps, silu):
    Cc = C1 + C2
    x = g(1, B, HW, Cc) * 2 + 0.5
    gamma, beta = 1 + 0.1 * g(2, Cc), 0.1 * g(3, Cc)
    x1 = x[..., :C1].contiguous()
    x2 = x[..., C1:].contiguous() if C2 else None
    out = run(rec, lambda: rec.groupnorm(h(x1), C1, h(x2) if C2 else None, C2, B, HW, G, eps, gamma.cuda(), beta.cuda(), silu))
    xr = x.half().float().permute(0, 2, 1)
    ref = F.group_norm(xr, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    close(out, ref.permute(0, 2, 1), what="groupnorm")


@pytest.mark.parametrize("B,HW,K,N,G", [(2, 256, 128, 320, 32), (2, 512, 64, 128, 32), (1, 1024, 192, 640, 32)])
def test_groupnorm_with_stats_fused_into_gemm_epilogue(rec, B, HW, K, N, G):
    """The producing GEMM emits per-channel (sum, sumsq) partials of its fp16 output; GroupNorm consumes them without
    re-reading the tensor.  Also covers the concat of a fused-stats tensor with a plain one."""
    if os.environ.get("BC_GEMM_TILE") or os.environ.get("BC_GEMM_GENERIC"):
        pytest.skip("fused statistics need the planned tile height")
    M = B * HW
    A, W, b, R = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N), g(4, M, N)
    gamma, beta = 1 + 0.1 * g(5, N), 0.1 * g(6, N)

    def fn():
        y = rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), R=h(R), ldr=N, rows_per_batch=HW,
                     want_gn=True)
        assert y.data_ptr() in rec.tots, "GEMM did not take the fused-statistics path"
        return y, rec.groupnorm(y, N, None, 0, B, HW, G, 1e-5, gamma.cuda(), beta.cuda(), True)
    y, out = run(rec, fn)
    assert rec.seg.kinds.get("groupnorm_fused_stats") == 1
    yref = A.half().float() @ W.half().float().t() + b + R.half().float()
    close(y, yref, what="gemm out")
    ref = F.silu(F.group_norm(y.float().cpu().view(B, HW, N).permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1)
    close(out, ref, what="groupnorm fused stats")
    # concat with a tensor that has no partials (stand-alone statistics pass for that source only)
    x2 = g(7, B, HW, 64)
    g2, b2 = 1 + 0.1 * g(8, N + 64), 0.1 * g(9, N + 64)
    out2 = run(rec, lambda: rec.groupnorm(y, N, h(x2), 64, B, HW, G, 1e-5, g2.cuda(), b2.cuda(), False))
    cat = torch.cat([y.float().cpu().view(B, HW, N), x2.half().float()], -1).permute(0, 2, 1)
    close(out2, F.group_norm(cat, G, g2, b2, 1e-5).permute(0, 2, 1), what="groupnorm concat mixed stats")


def test_groupnorm_stats_from_splitk_reducer(rec):
    """split-K GEMM: the vectorised reducer applies the epilogue and emits the GroupNorm partials (32-row slabs)."""
    B, HW, K, N, G = 2, 128, 1280, 1280, 32
    M = B * HW
    A, W, b = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N)
    gamma, beta = 1 + 0.1 * g(5, N), 0.1 * g(6, N)

    def fn():
        y = rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), rows_per_batch=HW, want_gn=True,
                     splitk=4)
        assert y.data_ptr() in rec.tots
        return y, rec.groupnorm(y, N, None, 0, B, HW, G, 1e-5, gamma.cuda(), beta.cuda(), True)
    y, out = run(rec, fn)
    assert rec.seg.kinds.get("groupnorm_fused_stats") == 1
    close(y, A.half().float() @ W.half().float().t() + b, what="splitk gemm out")
    ref = F.silu(F.group_norm(y.float().cpu().view(B, HW, N).permute(0, 2, 1), G, gamma, beta, 1e-5)).permute(0, 2, 1)
    close(out, ref, what="groupnorm stats from split-K reducer")


@pytest.mark.parametrize("HW", [1024, 4096, 8192])
def test_groupnorm_totals_poison_survives_many_addends(rec, HW):
    """ADVICE r4 / VERDICT r4 item 7 iv: a channel that is non-finite in EVERY producer workgroup (the common overflow case) must
    surface as NaN over its whole group - the reference's GroupNorm propagates NaN to the group - and not as finite garbage.  The
    round-4 poison (2^60 in slice 2) cancelled itself mod 2^64 after 16 addends; HW = 4096 on 128-row tiles is 32 addends."""
    from blobctrl_amd.launch import decode_gn_tot
    B, K, N, G = 1, 64, 320, 32
    M = B * HW
    A, W, b = g(1, M, K), g(2, N, K) / math.sqrt(K), g(3, N)
    b[7] = float("inf")                                         # channel 7 -> group 0 (channels 0..9) of the output
    gamma, beta = 1 + 0.1 * g(5, N), 0.1 * g(6, N)

    def fn():
        y = rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), rows_per_batch=HW, want_gn=True)
        tot = rec.tots[y.data_ptr()]
        return y, tot, rec.groupnorm(y, N, None, 0, B, HW, G, 1e-5, gamma.cuda(), beta.cuda(), True)
    y, tot, out = run(rec, fn)
    assert rec.seg.kinds.get("groupnorm_fused_stats") == 1
    # (N = 320: channel 7's partials go into the ten slots of the block of channels 0 .. 9, one per producer row tile - bc_gn_cg)
    addends = int((tot.cpu()[0, :10, 2] >> 50).sum())
    print(f"HW {HW}: {addends} poisoned addends on the block of channel 7")
    assert addends >= 4 and torch.isnan(decode_gn_tot(tot)[0, :10]).any()
    o = out.float().cpu().view(HW, N)
    assert torch.isnan(o[:, :10]).all(), "the poisoned group must read as NaN"
    assert torch.isfinite(o[:, 10:]).all(), "the other groups are untouched"
    # a standalone statistics pass over a tensor with a NaN channel, consumed by the convolution's in-prologue finalize
    x = g(7, B, HW, 64)
    x[:, :, 3] = float("nan")
    ab = run(rec, lambda: rec.gn_affine(h(x), 64, None, 0, B, HW, 8, 1e-5, torch.ones(64).cuda(), torch.zeros(64).cuda()))
    ab = ab.cpu()
    assert torch.isnan(ab[0, :8]).all() and torch.isfinite(ab[0, 8:]).all()


@pytest.mark.parametrize("k", [1, 8, 16, 32, 4096, 8191])
def test_groupnorm_totals_poison_by_addend_count(rec, k):
    """The same contract on hand-built totals: k poisoned addends on top of an honest sum, on either statistic, either sign of the
    wrapped slice - every consumer-side reader (bc_gn_tot_read) must see NaN."""
    from blobctrl_amd.launch import encode_gn_tot
    B, HW, Cc, G = 1, 4096, 64, 8
    x = g(9, B, HW, Cc)
    xs = x.half().float()
    base = encode_gn_tot(torch.stack([xs.sum(1), (xs * xs).sum(1)], -1))          # [B][C][6]
    for word in (2, 5):
        tot = base.clone()
        tot[0, 20, word] += k << 50                                                 # channel 20 -> group 2 (channels 16..23)
        t = tot.cuda()

        def fn():
            xd = h(x)
            rec.tots[xd.data_ptr()] = t
            return rec.gn_affine(xd, Cc, None, 0, B, HW, G, 1e-5, torch.ones(Cc).cuda(), torch.zeros(Cc).cuda())
        ab = run(rec, fn).cpu()
        assert torch.isnan(ab[0, 16:24]).all(), (k, word)
        assert torch.isfinite(ab[0, :16]).all() and torch.isfinite(ab[0, 24:]).all()


@pytest.mark.parametrize("rows,Cc", [(100, 320), (77, 640), (513, 1280), (9, 64), (5, 16)])
def test_layernorm(rec, rows, Cc):
    x, gamma, beta = g(1, rows, Cc) * 3 + 1, 1 + 0.1 * g(2, Cc), 0.1 * g(3, Cc)
    out = run(rec, lambda: rec.layernorm(h(x), rows, Cc, gamma.cuda(), beta.cuda(), 1e-5))
    close(out, F.layer_norm(x.half().float(), (Cc,), gamma, beta, 1e-5), what="layernorm")


# ---------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,heads,d,Nq,Nkv", [(2, 8, 40, 512, 512), (1, 8, 80, 256, 256), (2, 8, 160, 128, 128),
                                              (2, 8, 40, 300, 77), (1, 8, 160, 128, 77), (2, 4, 64, 257, 257),
                                              (2, 2, 8, 128, 128), (1, 2, 16, 32, 7), (1, 2, 32, 8, 8), (1, 8, 80, 200, 77),
                                              # D = 160 at the 16 x 32 level's 512 tokens, a ragged query block, fewer keys than one tile, ragged
                                              # last tiles
                                              (2, 8, 160, 512, 512), (1, 8, 160, 77, 512), (1, 2, 160, 40, 5), (1, 2, 160, 64, 1000),
                                              (1, 2, 160, 64, 1100)])
def test_attention(rec, B, heads, d, Nq, Nkv):
    Cc = heads * d
    q, k, v = g(1, B, Nq, Cc), g(2, B, Nkv, Cc), g(3, B, Nkv, Cc)
    ldvt = (Nkv + 63) // 64 * 64
    vt = torch.zeros(B, Cc, ldvt, dtype=torch.float16)
    vt[:, :, :Nkv] = v.half().transpose(1, 2)
    scale = d ** -0.5
    out = run(rec, lambda: rec.attention(h(q), h(k), vt.cuda(), rec.empty(B, Nq, Cc), B, heads, d, Nq, Nkv, Cc, Cc, ldvt, Cc,
                                         Nq * Cc, Nkv * Cc, Cc * ldvt, Nq * Cc, scale))
    qf = q.half().float().view(B, Nq, heads, d).transpose(1, 2)
    kf = k.half().float().view(B, Nkv, heads, d).transpose(1, 2)
    vf = v.half().float().view(B, Nkv, heads, d).transpose(1, 2)
    ref = torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf
    ref = ref.transpose(1, 2).reshape(B, Nq, Cc)
    close(out, ref, rtol=2e-3, atol=3e-3, what=f"attention d={d} Nq={Nq} Nkv={Nkv}")


@pytest.mark.parametrize("eight_waves", [True, False])
def test_attention_large_ragged_uses_128_vgpr_build(rec, eight_waves, monkeypatch):
    """B=2, 8 heads, d=40, 8450 tokens (a 65 x 130 canvas): the grid is large enough for the 128-VGPR (4 waves per SIMD) builds -
    the 8-wave workgroup (256 queries) the loop uses, and the 4-wave one (BC_ATTN_NO8) - and 8450 % 64 != 0 sends the last key tile
    through the masked tail (the path where those builds keep their few register spills); 8450 % 256 != 0 leaves a ragged query block."""
    if not eight_waves:
        monkeypatch.setenv("BC_ATTN_NO8", "1")
    B, heads, d, N = 2, 8, 40, 8450
    Cc = heads * d
    q, k, v = (g(s_, B, N, Cc).half().cuda() for s_ in (1, 2, 3))
    ldvt = (N + 63) // 64 * 64
    vt = torch.zeros(B, Cc, ldvt, dtype=torch.float16, device="cuda")
    vt[:, :, :N] = v.transpose(1, 2)
    out = run(rec, lambda: rec.attention(q, k, vt, rec.empty(B, N, Cc), B, heads, d, N, N, Cc, Cc, ldvt, Cc,
                                         N * Cc, N * Cc, Cc * ldvt, N * Cc, d ** -0.5))
    worst = 0.0
    for b in range(B):                                   # fp32 reference on the GPU, one (batch, head) at a time
        for hh in range(heads):
            sl = slice(hh * d, (hh + 1) * d)
            qf, kf, vf = q[b, :, sl].float(), k[b, :, sl].float(), v[b, :, sl].float()
            ref = torch.softmax(qf @ kf.t() * d ** -0.5, -1) @ vf
            worst = max(worst, float((out[b, :, sl].float() - ref).abs().max()))
    assert worst < 2e-3, f"max abs err {worst:.3e}"


def test_attention_d160_key_ranges_with_very_different_maxima(rec):
    """D = 160: key ranges whose score maxima differ by tens of log2 units, the larger ones late in the sequence (the running reference
    moves twice).  (Scores stay below ~25: the scale rides in the fp16 query, so a score's absolute error grows with its size.)"""
    B, heads, d, N = 1, 2, 160, 256
    Cc = heads * d
    q, k, v = g(1, B, N, Cc), g(2, B, N, Cc), g(3, B, N, Cc)
    k[:, 32:64] *= 3.0
    k[:, 224:256] *= 5.0
    ldvt = 256
    vt = torch.zeros(B, Cc, ldvt, dtype=torch.float16)
    vt[:, :, :N] = v.half().transpose(1, 2)
    out = run(rec, lambda: rec.attention(h(q), h(k), vt.cuda(), rec.empty(B, N, Cc), B, heads, d, N, N, Cc, Cc, ldvt, Cc,
                                         N * Cc, N * Cc, Cc * ldvt, N * Cc, d ** -0.5))
    qf = q.half().float().view(B, N, heads, d).transpose(1, 2)
    kf = k.half().float().view(B, N, heads, d).transpose(1, 2)
    vf = v.half().float().view(B, N, heads, d).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, N, Cc)
    close(out, ref, rtol=2e-3, atol=3e-3, what="attention d=160 with uneven key ranges")


def test_attention_online_softmax_rescale(rec):
    """Force the running-max rescale: a late key tile carries a much larger score than the early ones."""
    B, heads, d, N = 1, 2, 40, 256
    Cc = heads * d
    q, k, v = g(1, B, N, Cc), g(2, B, N, Cc), g(3, B, N, Cc)
    k[:, 200] = q[:, 10] * 4.0          # spike: query 10 suddenly matches key 200 (third tile)
    ldvt = N
    vt = v.half().transpose(1, 2).contiguous()
    out = run(rec, lambda: rec.attention(h(q), h(k), vt.cuda(), rec.empty(B, N, Cc), B, heads, d, N, N, Cc, Cc, ldvt, Cc,
                                         N * Cc, N * Cc, Cc * ldvt, N * Cc, d ** -0.5))
    qf = q.half().float().view(B, N, heads, d).transpose(1, 2)
    kf = k.half().float().view(B, N, heads, d).transpose(1, 2)
    vf = v.half().float().view(B, N, heads, d).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, N, Cc)
    close(out, ref, rtol=2e-3, atol=3e-3, what="attention rescale")


# ---------------------------------------------------------------------------------------------------- blob maths / glue
def test_splat_against_reference_fixtures(golden_dir):
    from blobctrl_amd.splat import blob_dict_from_ellipse, splat_features
    z = np.load(os.path.join(golden_dir, "splat.npz"))
    meta = json.loads(str(z["meta"]))
    for i, m in enumerate(meta):
        blob = blob_dict_from_ellipse(m["ellipse"], m["W"], m["H"])
        s = splat_features(**blob, score_size=(m["h"], m["w"]), return_d_score=True).cpu().numpy()
        assert s.dtype == np.float64 and s.shape == z[f"score_{i}"].shape
        np.testing.assert_allclose(s, z[f"score_{i}"], rtol=1e-9, atol=1e-12)
    blob = blob_dict_from_ellipse(meta[0]["ellipse"], 512, 512)
    blob["sizes"] = torch.tensor([[0.2]])
    s = splat_features(**blob, score_size=(64, 64), return_d_score=True).cpu().numpy()
    np.testing.assert_allclose(s, z["score_absent"], rtol=1e-12)
    with pytest.raises(NotImplementedError):
        splat_features(**blob, score_size=64, return_d_score=True)


def test_assemble_matches_construct_blobnet_input(rec):
    from oracle.pipeline import construct_input
    B, hh, ww, Fd = 2, 8, 8, 9
    lat, img, sc, feat = g(1, B, 4, hh, ww), g(2, 1, 4, hh, ww), g(3, 1, 1, hh, ww).abs(), g(4, 1, 1, Fd)
    Cpad = 16
    X = rec.zeros(2 * B, hh * 2 * ww, Cpad)
    dl, di, ds, df = lat.cuda(), img.cuda(), sc.cuda(), feat.cuda()
    run(rec, lambda: rec.call("bc_assemble_input", dl.data_ptr(), B, di.data_ptr(), ds.data_ptr(), df.data_ptr(), 1, Fd,
                              2 * B, hh, ww, Cpad, 0, X.data_ptr()))
    lmi = torch.cat([lat] * 2)
    feats = torch.einsum("nmhw,nmc->nchw", sc.repeat(2 * B, 1, 1, 1), feat.repeat(2 * B, 1, 1))
    ref = construct_input(lmi, sc.repeat(2 * B, 1, 1, 1), img.repeat(2 * B, 1, 1, 1), feats)
    got = from_nhwc(X, 2 * B, hh, 2 * ww)
    close(got[:, :4 + 1 + Fd], ref, rtol=1e-3, atol=2e-3, what="assemble")
    assert float(got[:, 4 + 1 + Fd:].abs().max()) == 0.0


def test_timestep_embedding(rec):
    from oracle.nets import timestep_embedding
    for t in (999.0, 981.0, 500.0, 1.0):
        out = rec.empty(2, 320)
        run(rec, lambda: rec.call("bc_timestep_embedding", None, None, t, 2, 320, out.data_ptr()))
        ref = timestep_embedding(torch.tensor([t, t]), 320)
        close(out, ref, rtol=0, atol=2e-3, what=f"temb t={t}")


@pytest.mark.parametrize("kind,n", [("unipc", 6), ("ddim", 5)])
def test_cfg_scheduler_step_matches_reference_trajectory(rec, golden_dir, kind, n):
    from blobctrl_amd.schedulers import DDIMTable, UniPCTable
    from oracle.schedulers import DDIMOracle, UniPCOracle
    B, hh, ww = 1, 8, 8
    tab = (UniPCTable() if kind == "unipc" else DDIMTable()).set_timesteps(n)
    orc = UniPCOracle() if kind == "unipc" else DDIMOracle()
    orc.set_timesteps(n)
    x = g(21, B, 4, hh, ww)
    lat = x.clone().cuda()
    coef = tab.table().cuda()
    idx = torch.zeros(1, dtype=torch.int32).cuda()
    hist = torch.zeros(3, B * 4 * hh * ww).cuda()
    eg = torch.zeros(B, 4, hh, ww).cuda()
    lib = rec.lib
    for i in range(n):
        eu, ec = g(200 + i, B, 4, hh, ww), g(300 + i, B, 4, hh, ww)
        full = torch.zeros(2 * B, hh, 2 * ww, 4)
        full[:B, :, ww:, :] = eu.permute(0, 2, 3, 1)
        full[B:, :, ww:, :] = ec.permute(0, 2, 3, 1)
        full[:, :, :ww, :] = 123.0                                    # left half must be ignored (pipe:1092-1093)
        fd = full.cuda()
        rc = lib.bc_cfg_scheduler_step(fd.data_ptr(), lat.data_ptr(), coef.data_ptr(), idx.data_ptr(), hist.data_ptr(), 7.5,
                                       B, hh, ww, eg.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        e = eu + 7.5 * (ec - eu)
        x = orc.step(e, x)
        torch.cuda.synchronize()
        close(eg, e, rtol=1e-5, atol=1e-5, what="cfg")
        close(lat, x, rtol=2e-5, atol=2e-5 * float(x.abs().max()), what=f"{kind} step {i}")
    assert int(idx.item()) == n


def test_layout_roundtrip(rec):
    B, Cc, HW = 2, 5, 24
    x = g(1, B, Cc, HW)
    lib = rec.lib
    s = torch.cuda.current_stream().cuda_stream
    xd = x.cuda()
    t = torch.empty(B, HW, 8, dtype=torch.float16, device="cuda")
    assert lib.bc_nchw_to_nhwc_f16(xd.data_ptr(), 1, B, Cc, HW, 8, t.data_ptr(), s) == 0
    back = torch.empty(B, Cc, HW, dtype=torch.float32, device="cuda")
    assert lib.bc_nhwc_to_nchw(t.data_ptr(), B, Cc, HW, 8, back.data_ptr(), 1, s) == 0
    torch.cuda.synchronize()
    close(back, x.half().float(), rtol=0, atol=0, what="layout roundtrip")
    assert float(t[..., Cc:].abs().max()) == 0.0


def test_assemble_im2col_is_the_unfolded_eight_channel_input(rec):
    """bc_assemble_input_im2col = F.unfold(3x3, pad 1) of what bc_assemble_input writes for 8 channels (4 latents, score, [score]),
    k = tap * 8 + channel, zero-filled from k = 72: the operand of conv_in as a dense K = 128 GEMM."""
    B, Bi, hh, ww = 2, 1, 6, 5
    lat, img, sc = g(1, B, 4, hh, ww), g(2, Bi, 4, hh, ww), torch.rand(Bi, hh, ww, generator=torch.Generator().manual_seed(3))
    dl, di, ds = lat.cuda(), img.cuda(), sc.cuda()
    for dup in (0, 1):
        x8 = rec.zeros(2 * B, hh * 2 * ww, 8)
        xi = rec.zeros(2 * B, hh * 2 * ww, 128)
        run(rec, lambda: (rec.call("bc_assemble_input", dl.data_ptr(), B, di.data_ptr(), ds.data_ptr(), None, Bi, 0, 2 * B, hh, ww, 8, dup,
                                   x8.data_ptr(), kind="assemble"),
                          rec.call("bc_assemble_input_im2col", dl.data_ptr(), B, di.data_ptr(), ds.data_ptr(), Bi, 2 * B, hh, ww, dup,
                                   xi.data_ptr(), kind="assemble")))
        nchw = x8.float().cpu().view(2 * B, hh, 2 * ww, 8).permute(0, 3, 1, 2)
        unf = F.unfold(nchw, 3, padding=1).view(2 * B, 8, 9, hh * 2 * ww).permute(0, 3, 2, 1).reshape(2 * B, hh * 2 * ww, 72)
        got = xi.float().cpu()
        assert torch.equal(got[..., :72], unf) and float(got[..., 72:].abs().max()) == 0.0
        assert float(got[..., 5].abs().max()) == (0.0 if not dup else float(got[..., 5].abs().max())) and (dup or float(nchw[:, 5].abs().max()) == 0.0)


def test_bad_arguments_fail_loudly(rec):
    from blobctrl_amd import _lib
    with pytest.raises(_lib.BlobCtrlHipError):
        run(rec, lambda: rec.gemm(A=h(g(1, 8, 12)), W=h(g(2, 8, 12)), M=8, N=8, K=12, out=rec.empty(8, 8)))   # K % 8
    with pytest.raises(_lib.BlobCtrlHipError):
        run(rec, lambda: rec.attention(h(g(1, 1, 8, 24)), h(g(2, 1, 8, 24)), h(g(3, 1, 24, 64)), rec.empty(1, 8, 24), 1, 2, 12,
                                       8, 8, 24, 24, 64, 24, 192, 192, 24 * 64, 192, 1.0))                       # d = 12
    # per-image weights / the softmax epilogue exist on the BC_TILE_GW* kernels only, and a softmax group is one 64 x 128 workgroup
    A, W = h(g(1, 128, 640)), h(g(2, 256, 640))
    with pytest.raises(_lib.BlobCtrlHipError):
        run(rec, lambda: rec.gemm(A=A, W=W, M=128, N=256, K=640, out=rec.empty(128, 256), w_bstride=256 * 640, rows_per_batch=64))
    with pytest.raises(_lib.BlobCtrlHipError):
        run(rec, lambda: rec.gemm(A=A, W=W, M=128, N=256, K=640, out=rec.empty(128, 160), ldc=160, tile_cfg=_lib.TILE_GW64x256, sm_group=128,
                                  sm_valid=77, sm_keep=80))
    with pytest.raises(_lib.BlobCtrlHipError):       # T > kept columns
        run(rec, lambda: rec.gemm(A=A, W=W, M=128, N=256, K=640, out=rec.empty(128, 160), ldc=160, tile_cfg=_lib.TILE_GW64x128, sm_group=128,
                                  sm_valid=90, sm_keep=80))
    lib = _lib.load()
    z = torch.zeros(16, dtype=torch.float16, device="cuda:0")
    assert lib.bc_ctx_fold(z.data_ptr(), 1280, z.data_ptr(), 128, 1, 81, 1280, 8, 1.0, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                           z.data_ptr(), z.data_ptr(), z.data_ptr(), None) != 0 and b"T <= 80" in lib.bc_last_error()


@pytest.mark.parametrize("kind", ["linear", "conv3x3"])
def test_lora_merged_equals_runtime_low_rank_branch_full_size(rec, kind):
    """the UNet LoRA is merged at load (W + (alpha / r) B A, weights.merge_lora; peft itself is absent here).  At the
    production shapes the merged layer must equal the adapter as peft EXECUTES it - base layer + scale * lora_B(lora_A(x)) as two
    extra small GEMMs at run time - for a Linear (to_q of the 64 x 128 level, M = 16384) and a Conv2d pair (3x3 rank-r conv then
    1x1, D/loaders/unet.py:314-340)."""
    from blobctrl_amd.weights import merge_lora, pack_conv3x3
    r, alpha = 4, 8.0
    scale = alpha / r
    if kind == "linear":
        M, C = 16384, 320
        x, w = g(1, M, C), g(2, C, C) / math.sqrt(C)
        a, b = g(3, r, C) / math.sqrt(C), g(4, C, r) * 0.5
        sd = merge_lora({"m.weight": w}, {"m.lora_A.weight": a, "m.lora_B.weight": b}, {"m": alpha})

        def fn():
            xd = h(x)
            merged = rec.gemm(A=xd, W=h(sd["m.weight"]), M=M, N=C, K=C, out=rec.empty(M, C))
            base = rec.gemm(A=xd, W=h(w), M=M, N=C, K=C, out=rec.empty(M, C))
            ap = torch.zeros(8, C)
            ap[:r] = a                                                       # rank padded to the 8-element K granularity
            bp = torch.zeros(C, 8)
            bp[:, :r] = b
            low = rec.gemm(A=xd, W=h(ap), M=M, N=8, K=C, out=rec.empty(M, 8))
            branch = rec.gemm(A=low, W=h(bp), M=M, N=C, K=8, out=rec.empty(M, C), alpha=scale, R=base, ldr=C)
            return merged, branch
        merged, branch = run(rec, fn)
        ref = x.half().float() @ (w + scale * b @ a).t()
    else:
        B, H, W, C = 2, 32, 64, 640
        x, w = g(1, B, C, H, W), g(2, C, C, 3, 3) / math.sqrt(9 * C)
        a, b = g(3, r, C, 3, 3) / math.sqrt(9 * C), g(4, C, r, 1, 1) * 0.5
        sd = merge_lora({"m.weight": w}, {"m.lora_A.weight": a, "m.lora_B.weight": b}, {"m": alpha})
        M = B * H * W
        conv = dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1)

        def fn():
            xd = h(x.permute(0, 2, 3, 1).reshape(M, C).contiguous())
            merged = rec.gemm(A=xd, W=h(pack_conv3x3(sd["m.weight"])), M=M, N=C, K=9 * C, out=rec.empty(M, C), conv=conv)
            base = rec.gemm(A=xd, W=h(pack_conv3x3(w)), M=M, N=C, K=9 * C, out=rec.empty(M, C), conv=conv)
            ap = torch.zeros(8, C, 3, 3)
            ap[:r] = a
            bp = torch.zeros(C, 8)
            bp[:, :r] = b.flatten(1)
            low = rec.gemm(A=xd, W=h(pack_conv3x3(ap)), M=M, N=8, K=9 * C, out=rec.empty(M, 8), conv=conv)
            branch = rec.gemm(A=low, W=h(bp), M=M, N=C, K=8, out=rec.empty(M, C), alpha=scale, R=base, ldr=C)
            return merged, branch
        merged, branch = run(rec, fn)
        ref = F.conv2d(x.half().float(), w + scale * (b.flatten(1) @ a.flatten(1)).reshape(w.shape), padding=1)
        ref = ref.permute(0, 2, 3, 1).reshape(M, C)
    close(merged, ref, rtol=4e-3, what=f"merged LoRA {kind} vs fp32")
    close(branch, ref, rtol=6e-3, what=f"runtime LoRA branch {kind} vs fp32")
    close(merged, branch, rtol=8e-3, what=f"merged vs runtime branch {kind}")


@pytest.mark.parametrize("tag", ["noalpha", "alpha"])
def test_lora_merged_layers_match_reference_lora_fixture(rec, golden_dir, tag):
    """f2 on the GPU: the merged Linear / Conv2d 3x3 of tests/golden/lora_merge.npz (outputs of the reference's own LoRA layers,
    D/models/lora.py) through bc_gemm - channels padded to the kernels' 8-element granularity."""
    from blobctrl_amd.weights import merge_lora, pack_conv3x3, pack_matrix
    z = np.load(os.path.join(golden_dir, "lora_merge.npz"))
    for kind in ("lin", "conv"):
        t = lambda n: torch.from_numpy(z[f"{kind}_{tag}_{n}"])
        rank = t("down").shape[0]
        alpha = float(z["network_alpha"]) if tag == "alpha" else float(rank)
        sd = merge_lora({"m.weight": t("w")}, {"m.lora_A.weight": t("down"), "m.lora_B.weight": t("up")}, {"m": alpha},
                        adapter_scale=float(z["lora_scale"]))
        x, ref = t("x"), t("y_fused")
        if kind == "lin":
            M, K, N = x.shape[0] * x.shape[1], x.shape[2], ref.shape[-1]
            out = run(rec, lambda: rec.gemm(A=h(x.reshape(M, K)), W=h(pack_matrix(sd["m.weight"])), M=M, N=N, K=K,
                                            out=rec.empty(M, N), bias=t("b").cuda()))
            close(out, ref.reshape(M, N), rtol=4e-3, what=f"merged LoRA linear ({tag}) vs the reference layer")
        else:
            B, C, H, W = x.shape
            Cp, N, M = 8, ref.shape[1], B * H * W
            xp = torch.zeros(B, H, W, Cp)
            xp[..., :C] = x.permute(0, 2, 3, 1)
            out = run(rec, lambda: rec.gemm(A=h(xp.reshape(M, Cp)), W=h(pack_conv3x3(sd["m.weight"])), M=M, N=N, K=9 * Cp,
                                            out=rec.empty(M, N), bias=t("b").cuda(),
                                            conv=dict(Cin=Cp, Hin=H, Win=W, Hout=H, Wout=W, stride=1)))
            close(out, ref.permute(0, 2, 3, 1).reshape(M, N), rtol=4e-3, what=f"merged LoRA conv3x3 ({tag}) vs the reference layer")


def test_run_timed_kernels_divides_a_splitk_gemm(rec):
    """bc_plan_run_timed_kernels: per-launch HIP-event times with a split-K GEMM divided into main kernel and reducer (what
    bench.py's roofline uses so that its per-kernel durations are comparable with rocprofv3's)."""
    M, N, K = 256, 1280, 11520
    A, W = g(1, M, K), g(2, N, K) / math.sqrt(K)
    seg = rec.begin("timed")
    rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), splitk=6)
    rec.gemm(A=h(A), W=h(W), M=M, N=N, K=K, out=rec.empty(M, N), splitk=1)
    s = torch.cuda.current_stream().cuda_stream
    seg.run(s)
    torch.cuda.synchronize()
    rows = seg.run_timed_kernels(s)
    whole = seg.run_timed(s)
    assert len(rows) == len(whole) == 2
    (_, main0, red0), (_, main1, red1) = rows
    assert main0 > 0 and red0 > 0 and main1 > 0 and red1 == 0.0
    assert abs((main0 + red0) - whole[0][1]) < 0.5 * whole[0][1] + 0.05      # same launches, separately timed replays


# ---------------------------------------------------------------------------------------------------- gemm_wreg.hip (BC_TILE_GW*)
def _gw_cfgs():
    from blobctrl_amd import _lib
    return [_lib.TILE_GW64x128, _lib.TILE_GW64x256, _lib.TILE_GW64x320]


@pytest.mark.parametrize("cfg_i", [0, 1, 2])
@pytest.mark.parametrize("mode", ["plain", "ln", "ln_geglu", "res_r2_gn", "concat", "alpha"])
def test_gemm_wreg_modes(rec, cfg_i, mode):
    """Small-M projections with the weights streamed into VGPRs against fp32 torch on the fp16-rounded operands: folded LayerNorm
    (raw rows in, statistics accumulated while staging), GEGLU, residual + BlobNet right-half residual + GroupNorm partials of the
    output, two-source A, per-image scale; every workgroup shape."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import fold_layernorm, pack_gemm_wreg
    cfg = _gw_cfgs()[cfg_i]
    nt = _lib.GW_TILES[cfg]
    B, rows = 2, 128
    M, K = B * rows, 640
    N = 64 * nt * (2 if mode != "ln_geglu" else 2)
    A = g(1, M, K) * 1.5 + 0.4
    W = g(2, N, K) / math.sqrt(K)
    b = g(3, N)
    x = A.half().float()
    kw, cs = {}, None
    wq, bq = W.half(), b
    if mode.startswith("ln"):
        gamma, beta = torch.rand(K, generator=torch.Generator().manual_seed(5)) + 0.5, g(6, K) * 0.1
        wq, cs, bq = fold_layernorm(W.half(), b, gamma, beta)
        x = F.layer_norm(x, (K,), gamma, beta, 1e-5)
        kw.update(ln_colsum=cs.cuda(), ln_eps=1e-5)
    ref = x @ W.half().float().t() + b
    n_out = N
    if mode == "ln_geglu":
        r4 = ref.view(M, N // 64, 2, 32)
        ref = (r4[:, :, 0] * F.gelu(r4[:, :, 1])).reshape(M, N // 2)
        n_out = N // 2
        kw.update(act=_lib.ACT_GEGLU)
    part = None
    if mode == "res_r2_gn":
        R, R2 = g(7, M, N), g(8, 1, rows, N)
        Wd, xmin = 16, 8                                            # canvas 8 x 16, residual added on the right half
        ref = ref + R.half().float()
        msk = (torch.arange(rows) % Wd >= xmin).float()[None, :, None]
        ref = (ref.view(B, rows, N) + R2.half().float() * msk).view(M, N)
        kw.update(R=h(R), ldr=N, R2=h(R2), ldr2=N, r2_xmin=xmin, r2_bmod=1, out_w=Wd, rows_per_batch=rows, want_gn=True)
    if mode == "alpha":
        tab = torch.tensor([9.0, 9.0, 0.7, 1.3]).cuda()
        idx = torch.tensor([1], dtype=torch.int32).cuda()
        ref = ref * 2.0 * torch.tensor([0.7, 1.3]).repeat_interleave(rows)[:, None]
        kw.update(alpha=2.0, alpha_dev=tab, alpha_idx=idx, alpha_bstride=B, rows_per_batch=rows)
    Ad = h(A)
    if mode == "concat":
        kw.update(A2=h(A[:, 320:].contiguous()), C1=320, lda=320, lda2=320)
        Ad = h(A[:, :320].contiguous())
    ws = pack_gemm_wreg(wq.cuda(), nt)
    out = run(rec, lambda: rec.gemm(A=Ad, W=ws, M=M, N=N, K=K, out=rec.empty(M, n_out), bias=bq.cuda(), tile_cfg=cfg, **kw))
    close(out, ref, what=f"gemm_wreg {mode} nt={nt}")
    if mode == "res_r2_gn":
        from blobctrl_amd.launch import decode_gn_tot, gn_tot_slots
        o = out.float().cpu().view(B, rows, N)
        want = torch.stack([o.sum(1), (o * o).sum(1)], -1)
        close(gn_tot_slots(decode_gn_tot(rec.tots[out.data_ptr()])), gn_tot_slots(want), rtol=1e-4, atol=1e-2, what="gemm_wreg GroupNorm statistics")


def test_gemm_wreg_folded_layernorm_with_a_large_row_mean(rec):
    """The folded LayerNorm takes its variance as E[x^2] - mean^2 from fp32 sums accumulated while the rows are staged: rows whose mean is
    12 standard deviations away from zero (far beyond what the residual streams of these blocks carry) still come out within 4e-3."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import fold_layernorm, pack_gemm_wreg
    M, K, N = 128, 1280, 256
    A = g(1, M, K) * 0.5 + 6.0
    W, b = g(2, N, K) / math.sqrt(K), g(3, N)
    gamma, beta = 1 + 0.2 * g(5, K), 0.1 * g(6, K)
    wq, cs, bq = fold_layernorm(W.half(), b, gamma, beta)
    out = run(rec, lambda: rec.gemm(A=h(A), W=pack_gemm_wreg(wq.cuda(), 2), M=M, N=N, K=K, out=rec.empty(M, N), bias=bq.cuda(),
                                    tile_cfg=_lib.TILE_GW64x128, ln_colsum=cs.cuda(), ln_eps=1e-5))
    ref = F.layer_norm(A.half().float(), (K,), gamma, beta, 1e-5) @ W.half().float().t() + b
    close(out, ref, rtol=4e-3, atol=4e-3 * float(ref.abs().max()), what="folded LayerNorm, mean = 12 sigma")


@pytest.mark.parametrize("cfg_i", [0, 2])
def test_gemm_wreg_groupnorm_in_the_row_staging(rec, cfg_i):
    """GroupNorm(x) -> 1x1 projection (Transformer2D norm -> proj_in) in ONE launch: the finalize runs in the kernel's prologue from the
    statistics totals of x (here from a bc_gn_stats pass), the affine is applied while the rows are staged."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_gemm_wreg
    cfg = _gw_cfgs()[cfg_i]
    nt = _lib.GW_TILES[cfg]
    B, HW, K, G = 2, 128, 640, 32
    N, M = 64 * nt * 2, B * HW
    x = g(1, B, HW, K) * 1.3 + 0.25 * g(2, 1, 1, K)
    W, b = g(3, N, K) / math.sqrt(K), g(4, N)
    gamma, beta = 1 + 0.2 * g(5, K), 0.2 * g(6, K)
    xd = h(x)
    out = run(rec, lambda: rec.gemm(A=xd, W=pack_gemm_wreg(W.half().cuda(), nt), M=M, N=N, K=K, out=rec.empty(M, N), bias=b.cuda(), tile_cfg=cfg,
                                    rows_per_batch=HW, a_gn=dict(x1=xd, C1=K, B=B, HW=HW, G=G, eps=1e-6, gamma=gamma.cuda(), beta=beta.cuda())))
    assert rec.seg.kinds.get("gn_stats") == 1 and "groupnorm" not in rec.seg.kinds
    y = F.group_norm(x.half().float().permute(0, 2, 1), G, gamma, beta, 1e-6).permute(0, 2, 1).reshape(M, K)
    close(out, y.half().float() @ W.half().float().t() + b, rtol=3e-3, what=f"gemm_wreg GroupNorm prologue nt={nt}")


@pytest.mark.parametrize("cfg_i", [0, 1])
def test_gemm_wreg_qkv_one_launch(rec, cfg_i):
    """LayerNorm -> to_q | to_k (row-major) and to_v (written transposed [B][C][ldvt] for the attention kernel) as ONE launch."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import fold_layernorm, pack_gemm_wreg
    cfg = _gw_cfgs()[cfg_i]
    nt = _lib.GW_TILES[cfg]
    B, rows, Cc = 2, 128, 512
    M = B * rows
    A, W = g(1, M, Cc) * 2 - 0.3, g(2, 3 * Cc, Cc) / math.sqrt(Cc)
    gamma, beta = torch.rand(Cc, generator=torch.Generator().manual_seed(5)) + 0.5, g(6, Cc) * 0.1
    Kp = 640                                                        # K must be a multiple of 320: pad the channels with zeros
    Ap = torch.zeros(M, Kp); Ap[:, :Cc] = A
    Wp = torch.zeros(3 * Cc, Kp); Wp[:, :Cc] = W
    gp, bp_ = torch.ones(Kp), torch.zeros(Kp); gp[:Cc], bp_[:Cc] = gamma, beta
    # (the padded LayerNorm is over 640 channels: the reference below is stated on the padded operands)
    x = F.layer_norm(Ap.half().float(), (Kp,), gp, bp_, 1e-5)
    ref = x @ Wp.half().float().t()
    wq, cs, bq = fold_layernorm(Wp.half(), None, gp, bp_)
    ldvt = 192
    qk, vt = None, None

    def go():
        nonlocal qk, vt
        qk, vt = rec.empty(M, 2 * Cc), rec.zeros(B, Cc, ldvt)
        rec.gemm(A=h(Ap), W=pack_gemm_wreg(wq.cuda(), nt), M=M, N=3 * Cc, K=Kp, out=qk, bias=bq.cuda(), tile_cfg=cfg, ln_colsum=cs.cuda(),
                 C_t=vt, ldc_t=ldvt, n_t0=2 * Cc, rows_per_batch=rows)
    run(rec, go)
    close(qk, ref[:, :2 * Cc], what="q | k")
    v = ref[:, 2 * Cc:].view(B, rows, Cc).permute(0, 2, 1)
    close(vt[:, :, :rows], v, what="V^T")
    assert float(vt[:, :, rows:].abs().max()) == 0.0               # the padding of the token axis is left alone


def test_gemm_wreg_pack_matches_host_packing(rec):
    """bc_gemm_wreg_pack (device) writes the same bytes as weights.pack_gemm_wreg (torch), for every configuration."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_gemm_wreg
    lib = _lib.load()
    for cfg, nt in _lib.GW_TILES.items():
        N, K = 64 * nt * 3, 320
        w = h(g(9, N, K))
        out = torch.zeros(lib.bc_gemm_wreg_stream_elems(N, K), dtype=torch.float16, device="cuda:0")
        _lib.check(lib.bc_gemm_wreg_pack(w.data_ptr(), K, N, K, cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "pack")
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), pack_gemm_wreg(w, nt).cpu())
    assert lib.bc_gemm_wreg_eligible(1024, 1280, 1280, 0, _lib.TILE_GW64x128) == 1
    assert lib.bc_gemm_wreg_eligible(1000, 1280, 1280, 0, _lib.TILE_GW64x128) == 0      # M % 64
    assert lib.bc_gemm_wreg_eligible(1024, 1280, 1024, 0, _lib.TILE_GW64x128) == 0      # K % 320
    assert lib.bc_gemm_wreg_eligible(1024, 1280, 2560, 1280, _lib.TILE_GW64x320) == 1


def _ctx_fold_case(B=2, T=77, Cc=1280, heads=8):
    """Seeded stand-ins for a block's projected prompt and its attn2 weights (SD-1.5's 1280-channel block: 8 heads of 160)."""
    k, v = g(31, B, T, Cc) * 1.2, g(32, B, T, Cc)
    ldvt = 128
    vt = torch.zeros(B, Cc, ldvt)
    vt[:, :, :T] = v.permute(0, 2, 1)
    Wq, Wo = g(33, Cc, Cc) / math.sqrt(Cc), g(34, Cc, Cc) / math.sqrt(Cc)
    gamma, beta = 1 + 0.2 * g(35, Cc), 0.1 * g(36, Cc)
    return k, vt, ldvt, Wq, Wo, gamma, beta


@pytest.mark.parametrize("B,T,Cc", [(2, 77, 1280), (1, 17, 640), (3, 80, 1280)])
def test_ctx_fold_matches_its_host_statement(rec, B, T, Cc):
    """bc_ctx_fold (the prompt folded into attn2's weights, once per edit) against weights.fold_cross_attention: the two fragment streams
    within one fp16 ulp of the host's fp32 products (the kernel sums the head width in index order), the column sums exactly those of the
    ROUNDED rows the kernel wrote, keys >= T zero."""
    from blobctrl_amd.weights import fold_cross_attention, fold_layernorm, pack_gemm_wreg
    heads = 8
    k, vt, ldvt, Wq, Wo, gamma, beta = _ctx_fold_case(B, T, Cc, heads)
    wq, _, bq = fold_layernorm(Wq.half(), None, gamma, beta)
    scale = (Cc // heads) ** -0.5
    got = None

    def go():
        nonlocal got
        got = rec.ctx_fold(h(k.reshape(B * T, Cc)), h(vt), B, T, Cc, ldvt, heads, scale, wq.cuda(), bq.cuda(), h(Wo))
    run(rec, go)
    wqk, cs, qb, vwo = got
    qk, cs_ref, qb_ref, vo = fold_cross_attention(k.half(), vt.half(), T, heads, scale, wq, bq, Wo.half())
    for b in range(B):
        want_qk, want_vo = pack_gemm_wreg(qk[b], 2).float(), pack_gemm_wreg(vo[b], 2).float()
        have_qk, have_vo = wqk[b].float().cpu(), vwo[b].float().cpu()
        assert have_qk.shape == want_qk.shape and have_vo.shape == want_vo.shape
        # one fp16 ulp of the entry (2^-10 relative) + the fp32 summation-order noise of near-cancelling entries
        assert float(((have_qk - want_qk).abs() - 1e-3 * want_qk.abs()).max()) < 2e-5, "QK stream"
        assert float(((have_vo - want_vo).abs() - 1e-3 * want_vo.abs()).max()) < 2e-4, "VO stream"
    close(qb, qb_ref, rtol=1e-4, atol=1e-5, what="bias row scale K . (Wq beta)")
    # column sums: of the rows the KERNEL rounded (its own bytes), in fp32
    rows_qk = torch.zeros(B, heads * 128, Cc)
    for b in range(B):
        v2 = wqk[b].float().cpu()[: heads * 128 * Cc].view(heads, 4, Cc // 32, 2, 4, 16, 8)     # [j, wave, s, t, q, r, e]
        rows_qk[b] = v2.permute(0, 1, 3, 5, 2, 4, 6).reshape(heads * 128, Cc)
    close(cs, rows_qk.sum(-1), rtol=1e-4, atol=1e-4, what="column sums of the rounded QK rows")
    pad = rows_qk.view(B, heads, 128, Cc)[:, :, T:]
    assert float(pad.abs().max()) == 0.0 and float(cs.view(B, heads, 128)[:, :, T:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,T,Cc", [(128, 77, 1280), (512, 77, 1280), (64, 33, 640)])
def test_cross_attention_as_two_folded_projections(rec, rows, T, Cc):
    """to_q + 77-key attention + to_out of a 1280-channel block as bc_ctx_fold (per edit) + two bc_gemm launches with per-image weights
    (softmax over the 128 padded keys of a head in the first one's epilogue, 80 probabilities per head kept) against fp32 torch: LayerNorm -> to_q -> per-head
    softmax(q k^T / sqrt(d)) v -> to_out + bias + residual (attention.py:491-510, attention_processor.py:2191-2224)."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import fold_layernorm
    B, heads = 2, 8
    D = Cc // heads
    M = B * rows
    k, vt, ldvt, Wq, Wo, gamma, beta = _ctx_fold_case(B, T, Cc, heads)
    x = g(41, M, Cc) * 1.5 + 0.3
    bo = g(42, Cc) * 0.2
    wq, _, bq = fold_layernorm(Wq.half(), None, gamma, beta)
    scale = D ** -0.5
    out = None

    def go():
        nonlocal out
        wqk, cs, qb, vwo = rec.ctx_fold(h(k.reshape(B * T, Cc)), h(vt), B, T, Cc, ldvt, heads, scale, wq.cuda(), bq.cuda(), h(Wo))
        N, Np = 128 * heads, 80 * heads
        xd = h(x)
        pr = rec.empty(M, Np)
        rec.gemm(A=xd, W=wqk, M=M, N=N, K=Cc, out=pr, ldc=Np, bias=qb, tile_cfg=_lib.TILE_GW64x128, ln_colsum=cs, rows_per_batch=rows,
                 w_bstride=wqk.shape[1], vec_bstride=N, sm_group=128, sm_valid=T, sm_keep=80)
        out = rec.empty(M, Cc)
        rec.gemm(A=pr, W=vwo, M=M, N=Cc, K=Np, out=out, bias=bo.cuda(), tile_cfg=_lib.TILE_GW64x128, R=xd, ldr=Cc, rows_per_batch=rows,
                 w_bstride=vwo.shape[1])
        go.pr = pr
    run(rec, go)
    xf = x.half().float()
    q = (F.layer_norm(xf, (Cc,), gamma, beta, 1e-5) @ Wq.half().float().t()).view(B, rows, heads, D).permute(0, 2, 1, 3)
    kh = k.half().float().view(B, T, heads, D).permute(0, 2, 1, 3)
    vh = vt.half().float()[:, :, :T].reshape(B, heads, D, T).permute(0, 1, 3, 2)
    p = torch.softmax(q @ kh.transpose(-1, -2) * scale, -1)
    a = (p @ vh).permute(0, 2, 1, 3).reshape(M, Cc)
    ref = a @ Wo.half().float().t() + bo + xf
    pr = go.pr.float().cpu().view(B, rows, heads, 80)
    assert float(pr[..., T:].abs().max()) == 0.0                       # padded keys carry no probability
    close(pr[..., :T].permute(0, 2, 1, 3), p, rtol=2e-2, atol=2e-3, what="probabilities")
    close(out, ref, what=f"folded cross-attention, {rows} rows per image")


# ---------------------------------------------------------------------------------------------------- BC_TILE_G256 (gemm256.hip, round 6)
@pytest.mark.parametrize("mode", ["plain", "geglu", "gelu_alpha", "res_r2_gn", "transposed", "rowscale", "qkv", "two_source"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 1280), (1024, 512, 256), (9728, 2560, 384)])
def test_gemm256_modes(rec, mode, M, N, K):
    """The 256 x 256 / 8-wave / 8-phase LDS-DMA GEMM against fp32 torch on the fp16-rounded operands: every epilogue mode the plan uses it
    with, at tile counts from 1 (one workgroup, two k-tiles: prologue + drain only) to 360 (persistent workgroups that take a second
    tile while the first one's epilogue runs; the 9 x 4-row grouped tile order with a ragged last group of 2 row tiles)."""
    from blobctrl_amd import _lib
    if M == 9728 and mode not in ("plain", "res_r2_gn", "geglu", "qkv", "two_source"):
        pytest.skip("the large case runs the modes that differ in the persistent staging / epilogue")
    if mode == "qkv" and N < 512:
        pytest.skip("q | k | V^T needs at least two column tiles")
    if mode == "two_source" and K < 256:
        pytest.skip("two sources need at least one k-tile pair each")
    B = 2 if M % 512 == 0 else 1
    rows = M // B
    A = g(1, M, K) * 1.3 + 0.2
    W = g(2, N, K) / math.sqrt(K)
    b = g(3, N)
    x = A.half().float()
    ref = x @ W.half().float().t() + b
    kw, n_out, out = {}, N, None
    if mode == "geglu":
        r4 = ref.view(M, N // 64, 2, 32)
        ref = (r4[:, :, 0] * F.gelu(r4[:, :, 1])).reshape(M, N // 2)
        n_out = N // 2
        kw.update(act=_lib.ACT_GEGLU)
    if mode == "gelu_alpha":
        ref = F.gelu(ref) * 0.7
        kw.update(act=_lib.ACT_GELU, alpha=0.7)
    if mode == "res_r2_gn":
        R, R2 = g(7, M, N), g(8, 1, rows, N)
        Wd, xmin = 16, 8                                            # canvas rows / 16 x 16, residual added on the right half
        ref = ref + R.half().float()
        msk = (torch.arange(rows) % Wd >= xmin).float()[None, :, None]
        ref = (ref.view(B, rows, N) + R2.half().float() * msk).view(M, N)
        kw.update(R=h(R), ldr=N, R2=h(R2), ldr2=N, r2_xmin=xmin, r2_bmod=1, out_w=Wd, rows_per_batch=rows, want_gn=rows % 256 == 0)
    if mode == "rowscale":
        tab = torch.tensor([9.0] * B + [0.7, 1.3][:B]).cuda()             # row 1 of a [steps][B] table
        idx = torch.tensor([1], dtype=torch.int32).cuda()
        sc = torch.tensor([0.7, 1.3])[:B].repeat_interleave(rows)[:, None]
        ref = ref * 2.0 * sc
        kw.update(alpha=2.0, alpha_dev=tab, alpha_idx=idx, alpha_bstride=B, rows_per_batch=rows)
    if mode == "transposed":
        ldc = rows + 64
        out = torch.zeros(B, N, ldc, dtype=torch.float16, device="cuda")
        kw.update(out_mode=_lib.OUT_F16_T, ldc=ldc, rows_per_batch=rows)
        ref = ref.view(B, rows, N).permute(0, 2, 1)
    vt, n_t0 = None, 0
    if mode == "qkv":                                               # the last 256 columns transposed into C_t, the others row-major (ldc = n_t0)
        n_t0, ldvt = N - 256, rows + 64
        vt = torch.zeros(B, 256, ldvt, dtype=torch.float16, device="cuda")
        kw.update(C_t=vt, ldc_t=ldvt, n_t0=n_t0, rows_per_batch=rows)
        n_out = n_t0
    Ad, c1 = h(A), 0
    if mode == "two_source":                                        # columns k >= C1 from a second tensor with its own row stride
        c1 = 128 if K == 256 else (K // 256) * 128
        Ad = h(A[:, :c1].contiguous())
        kw.update(A2=h(A[:, c1:].contiguous()), C1=c1, lda=c1, lda2=K - c1)
    assert rec.lib.bc_gemm256_eligible(M, N, K, c1, kw.get("out_mode", _lib.OUT_F16), kw.get("rows_per_batch", 0) or M, 1 if kw.get("want_gn") else 0)
    res = run(rec, lambda: rec.gemm(A=Ad, W=h(W), M=M, N=N, K=K, out=out if out is not None else rec.empty(M, n_out), bias=b.cuda(),
                                    tile_cfg=_lib.TILE_G256, **kw))
    assert rec.seg.meta[-1]["rocprof"] == "gemm256_kernel"
    if mode == "qkv":
        close(res, ref[:, :n_t0], what=f"gemm256 q | k part {M}x{N}x{K}")
        close(vt[:, :, :rows], ref[:, n_t0:].view(B, rows, 256).permute(0, 2, 1), what=f"gemm256 V^T part {M}x{N}x{K}")
        assert float(vt[:, :, rows:].abs().max()) == 0.0
        return
    if mode == "transposed":
        close(res[:, :, :rows], ref, what=f"gemm256 transposed {M}x{N}x{K}")
        assert float(res[:, :, rows:].abs().max()) == 0.0
    else:
        close(res, ref, what=f"gemm256 {mode} {M}x{N}x{K}")
    if kw.get("want_gn"):
        from blobctrl_amd.launch import decode_gn_tot, gn_tot_slots
        o = res.float().cpu().view(B, rows, N)
        want = torch.stack([o.sum(1), (o * o).sum(1)], -1)
        close(gn_tot_slots(decode_gn_tot(rec.tots[res.data_ptr()])), gn_tot_slots(want), rtol=1e-4, atol=1e-2 * max(1.0, rows / 256), what="gemm256 GroupNorm statistics")


def test_gemm256_replays_are_bit_identical_and_equal_the_lds_dma_tiles(rec):
    """Race screen for the counted-vmcnt / raw-barrier pipeline (a read placed one phase early passes whenever the DMA happens to land
    first): 12 replays of a 640-tile launch under changing cache state give the same bits, and those bits equal gemm_fast's 256 x 128
    tile (same MFMA accumulation order per output: k ascending in steps of 32 vs 16 - so equal to fp32 rounding, not bit for bit)."""
    from blobctrl_amd import _lib
    M, N, K = 4096, 10240, 1280
    A, W, b = g(11, M, K), g(12, N, K) / math.sqrt(K), g(13, N)
    Ad, Wd, bd = h(A), h(W), b.cuda()
    o1, o2 = rec.empty(M, N // 2), rec.empty(M, N // 2)
    seg = rec.begin("g256_replay")
    rec.gemm(A=Ad, W=Wd, M=M, N=N, K=K, out=o1, bias=bd, act=_lib.ACT_GEGLU, tile_cfg=_lib.TILE_G256)
    rec.gemm(A=Ad, W=Wd, M=M, N=N, K=K, out=o2, bias=bd, act=_lib.ACT_GEGLU, tile_cfg=1)
    s = torch.cuda.current_stream().cuda_stream
    seg.run(s)
    torch.cuda.synchronize()
    first = o1.clone()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for i in range(12):
        o1.zero_()
        if i % 3 == 0:
            junk.fill_(i)                                           # evict L2 / Infinity Cache: another arrival order of the half-tiles
        seg.run(s)
        torch.cuda.synchronize()
        assert torch.equal(o1, first), f"replay {i} differs"
    close(o1, o2, rtol=2e-3, what="gemm256 vs gemm_fast 256x128")
