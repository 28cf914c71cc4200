"""Parity of the LDS-resident input-halo convolution with the fused GroupNorm(+SiLU) prologue - both builds: weights by LDS-DMA
(conv_halo.hip, BC_TILE_HALO) and weights streamed into VGPRs from the packed fragment stream (conv_wreg.hip, BC_TILE_WREG) -
against a plain PyTorch fp32 statement of resnet.py:327-341 / 351-366 (torch.cat -> GroupNorm -> SiLU -> conv3x3 [+ bias, time
embedding row vector, residual, BlobNet right-half residual]) computed on the CPU from the same fp16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.common import g  # noqa: E402
from tests.test_kernels_gpu import close, h, rec, run  # noqa: E402,F401


@pytest.fixture(params=["halo", "wreg"])
def tile(request):
    return request.param


def tcfg(tile):
    from blobctrl_amd import _lib
    return _lib.TILE_WREG if tile.startswith("wreg") else _lib.TILE_HALO


def wmat(w, tile):
    """fp16 device weights of a 3x3 convolution in the layout the tile configuration reads."""
    from blobctrl_amd.weights import pack_conv3x3, pack_conv_wreg
    p = pack_conv3x3(w).half()
    return (pack_conv_wreg(p) if tile.startswith("wreg") else p).cuda()


def nhwc(x):            # [B,C,H,W] fp32 -> token-major fp16 on the GPU
    B, C, H, W = x.shape
    return h(x.permute(0, 2, 3, 1).reshape(B, H * W, C).contiguous())


def from_nhwc(t, B, H, W):
    return t.float().cpu().view(B, H, W, -1).permute(0, 3, 1, 2)


def conv_ref(x, w, b):
    return F.conv2d(x.half().float(), w.half().float(), b, padding=1)


@pytest.mark.parametrize("B,H,W,Cin,Cout,sk", [(1, 8, 16, 64, 160, 1), (2, 16, 32, 128, 320, 1), (1, 24, 16, 192, 160, 1),
                                               (2, 8, 16, 256, 160, 2), (1, 8, 32, 640, 320, 5), (1, 16, 16, 320, 160, None)])
def test_halo_conv_plain(rec, tile, B, H, W, Cin, Cout, sk):
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_conv3x3
    x, w, b = g(1, B, Cin, H, W), g(2, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(3, Cout)
    M = B * H * W
    out = run(rec, lambda: rec.gemm(A=nhwc(x), W=wmat(w, tile), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout), bias=b.cuda(),
                                    conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1), rows_per_batch=H * W,
                                    tile_cfg=tcfg(tile), splitk=sk))
    close(from_nhwc(out, B, H, W), conv_ref(x, w, b), what=f"halo conv {B}x{Cin}->{Cout}@{H}x{W} sk={sk}")


@pytest.mark.parametrize("finalize", ["launch", "in_kernel"])
@pytest.mark.parametrize("B,H,W,C1,C2,Cout,sk,silu", [(2, 16, 16, 128, 0, 160, 1, True), (1, 8, 32, 64, 128, 320, 1, True),
                                                       (2, 8, 16, 320, 320, 160, 2, True), (1, 16, 32, 128, 0, 160, 1, False),
                                                       (1, 8, 16, 640, 320, 160, 3, True),
                                                       # a 1280-channel span in ONE workgroup: wider than the counted-wait finalize of
                                                       # conv_wreg.hip takes (1024 channels) - its round-4 form, every wave, three barriers
                                                       (1, 8, 16, 1280, 0, 160, 1, True)])
def test_halo_conv_fused_groupnorm_concat_epilogue(rec, tile, B, H, W, C1, C2, Cout, sk, silu, finalize, monkeypatch):
    """GroupNorm statistics from a standalone pass -> bc_gn_finalize -> affine applied in the halo staging (zero padding AFTER the
    activation), two channel-concatenated sources, and the whole ResBlock epilogue: bias + time-embedding row vector + residual +
    BlobNet right-half residual + GroupNorm partials of the output."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_conv3x3
    # (conv_halo.hip finalizes in its prologue only for narrow spans: for the 1280-channel case the recorder falls back to a finalize launch there)
    halo_falls_back = tile == "halo" and C1 + C2 > 1024 and sk == 1
    Cin, G = C1 + C2, 8
    x1 = g(1, B, C1, H, W) * 1.7 + 0.3
    x2 = g(4, B, C2, H, W) * 0.6 - 0.2 if C2 else None
    w, b = g(2, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(3, Cout)
    gamma, beta = 1.0 + 0.2 * g(5, Cin), 0.3 * g(6, Cin)
    temb, R, R2 = g(7, B, Cout), g(8, B, Cout, H, W), g(9, 1, Cout, H, W)
    M, HW = B * H * W, H * W
    xcat = torch.cat([x1, x2], 1) if C2 else x1

    def fn():
        t1, t2 = nhwc(x1), (nhwc(x2) if C2 else None)
        if finalize == "launch":
            kw = dict(a_affine=rec.gn_affine(t1, C1, t2, C2, B, HW, G, 1e-5, gamma.cuda(), beta.cuda()))
        else:                                # GroupNorm finalize inside the convolution's prologue (no bc_gn_finalize launch)
            kw = dict(a_gn=dict(x1=t1, C1=C1, x2=t2, C2=C2, B=B, HW=HW, G=G, eps=1e-5, gamma=gamma.cuda(), beta=beta.cuda()))
        if C2:
            kw.update(A2=t2, C1=C1, lda2=C2)
        out = rec.gemm(A=t1, lda=C1, W=wmat(w, tile), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout), bias=b.cuda(),
                       conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1), rows_per_batch=HW, tile_cfg=tcfg(tile),
                       splitk=sk, a_act=_lib.ACT_SILU if silu else _lib.ACT_NONE, rowvec=h(temb), ld_rowvec=Cout,
                       R=nhwc(R), ldr=Cout, R2=nhwc(R2), ldr2=Cout, r2_xmin=W - H if W > H else 0, r2_bmod=1, out_w=W, want_gn=True,
                       **kw)
        if finalize == "in_kernel" and not halo_falls_back:
            assert "gnfin" in rec.seg.meta[-1]["variant"], rec.seg.meta[-1]["variant"]
        return out, rec.tots[out.data_ptr()]
    out, part = run(rec, fn)
    xh = xcat.half().float()
    y = F.group_norm(xh, G, gamma, beta, 1e-5)
    y = F.silu(y) if silu else y
    ref = F.conv2d(y.half().float(), w.half().float(), b, padding=1) + temb.half().float()[:, :, None, None] + R.half().float()
    xmin = W - H if W > H else 0
    ref[..., xmin:] += R2.half().float()[..., xmin:]
    got = from_nhwc(out, B, H, W)
    close(got, ref, rtol=4e-3, what=f"fused GN conv {C1}+{C2}->{Cout}@{H}x{W} sk={sk}")
    # GroupNorm statistics of the fp16-rounded output: the totals every workgroup added its rows to
    from blobctrl_amd.launch import decode_gn_tot, gn_tot_slots
    s = gn_tot_slots(decode_gn_tot(part).float())                             # [B][Cout / cg][2]: sums per totals block
    o = out.float().cpu().view(B, HW, Cout)
    want = gn_tot_slots(torch.stack([o.sum(1), (o * o).sum(1)], -1))
    assert torch.allclose(s[..., 0], want[..., 0], rtol=1e-3, atol=1e-2 * HW ** 0.5)
    assert torch.allclose(s[..., 1], want[..., 1], rtol=1e-3, atol=1e-2 * HW ** 0.5)


def test_halo_conv_matches_the_unfused_path_on_a_resblock_shape(rec, tile):
    """The production shape family: 320 -> 320 @ 64 x 128, batch 2, against GroupNorm pass + implicit-GEMM kernel (both HIP)."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_conv3x3
    B, H, W, C, G = 2, 64, 128, 320, 32
    x, w, b = g(1, B, C, H, W), g(2, C, C, 3, 3) / math.sqrt(9 * C), g(3, C)
    gamma, beta = 1.0 + 0.2 * g(5, C), 0.3 * g(6, C)
    M, HW = B * H * W, H * W
    conv = dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1)

    def fn():
        t = nhwc(x)
        ab = rec.gn_affine(t, C, None, 0, B, HW, G, 1e-5, gamma.cuda(), beta.cuda())
        fused = rec.gemm(A=t, lda=C, W=wmat(w, tile), M=M, N=C, K=9 * C, out=rec.empty(M, C), bias=b.cuda(), conv=conv,
                         rows_per_batch=HW, tile_cfg=tcfg(tile), a_affine=ab, a_act=_lib.ACT_SILU)
        y = rec.groupnorm(t, C, None, 0, B, HW, G, 1e-5, gamma.cuda(), beta.cuda(), True)
        plain = rec.gemm(A=y, W=h(pack_conv3x3(w)), M=M, N=C, K=9 * C, out=rec.empty(M, C), bias=b.cuda(), conv=conv,
                         rows_per_batch=HW)
        return fused, plain
    fused, plain = run(rec, fn)
    close(fused, plain, rtol=4e-3, what="fused vs unfused resblock conv")


def test_halo_conv_single_source_with_a_wider_pixel_stride(rec, tile):
    """`lda` is the PIXEL stride of A (blobctrl_hip.h): a single-source halo convolution over the first 128 channels of a
    192-channel NHWC buffer (lda = 192 > Cin = 128) must read that view, not a packed 128-channel image."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_conv3x3
    B, H, W, Cwide, Cin, Cout = 2, 8, 16, 192, 128, 160
    x, w, b = g(1, B, Cwide, H, W), g(2, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(3, Cout)
    M = B * H * W
    out = run(rec, lambda: rec.gemm(A=nhwc(x), lda=Cwide, W=wmat(w, tile), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout),
                                    bias=b.cuda(), conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1), rows_per_batch=H * W,
                                    tile_cfg=tcfg(tile)))
    close(from_nhwc(out, B, H, W), conv_ref(x[:, :Cin], w, b), what="halo conv over a strided single-source view")
    with pytest.raises(_lib.BlobCtrlHipError):
        run(rec, lambda: rec.gemm(A=nhwc(x), lda=64, W=wmat(w, tile), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout),
                                  conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1), rows_per_batch=H * W,
                                  tile_cfg=tcfg(tile)))


@pytest.mark.parametrize("N,Cin", [(160, 64), (320, 320), (1280, 128)])
def test_wreg_pack_kernel_equals_the_host_packing(N, Cin):
    """bc_conv_wreg_pack (device) and weights.pack_conv_wreg (host) write the same fragment streams."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_conv_wreg
    lib = _lib.load()
    w = (g(11, N, 9 * Cin) * 8).half()
    wd = w.cuda()
    out = torch.empty_like(wd)
    _lib.check(lib.bc_conv_wreg_pack(wd.data_ptr(), N, Cin, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "bc_conv_wreg_pack")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), pack_conv_wreg(w))


@pytest.mark.parametrize("B,H,W,Cin,Cout,sk", [(1, 4, 8, 64, 160, 1), (2, 8, 16, 128, 320, 1), (1, 16, 16, 256, 160, 2), (2, 8, 8, 1280, 160, None)])
def test_wreg_conv_behind_a_2x_nearest_upsample(rec, B, H, W, Cin, Cout, sk):
    """Upsample2D (D/models/upsampling.py: F.interpolate(scale_factor=2.0, mode="nearest") -> conv3x3) on BC_TILE_WREG: the halo rows
    are fetched from source pixel (y / 2, x / 2); the virtual image never exists."""
    from blobctrl_amd import _lib
    x, w, b = g(1, B, Cin, H, W), g(2, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(3, Cout)
    Hv, Wv = 2 * H, 2 * W
    M = B * Hv * Wv
    R2 = g(9, 1, Cout, Hv, Wv)
    out = run(rec, lambda: rec.gemm(A=nhwc(x), lda=Cin, W=wmat(w, "wreg"), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout), bias=b.cuda(),
                                    conv=dict(Cin=Cin, Hin=H, Win=W, Hv=Hv, Wv=Wv, Hout=Hv, Wout=Wv, stride=1), rows_per_batch=Hv * Wv,
                                    tile_cfg=_lib.TILE_WREG, splitk=sk, R2=nhwc(R2), ldr2=Cout, r2_xmin=0, r2_bmod=1, out_w=Wv))
    ref = F.conv2d(F.interpolate(x.half().float(), scale_factor=2.0, mode="nearest"), w.half().float(), b, padding=1) + R2.half().float()
    close(from_nhwc(out, B, Hv, Wv), ref, what=f"upsample + conv {B}x{Cin}->{Cout}@{H}x{W}->{Hv}x{Wv} sk={sk}")


@pytest.mark.parametrize("B,H,W,C1,C2,Cout", [(2, 32, 64, 320, 0, 320), (2, 8, 16, 1280, 1280, 1280), (1, 8, 16, 1280, 0, 1280)])
def test_conv_is_bit_reproducible_under_changing_cache_state(rec, tile, B, H, W, C1, C2, Cout):
    """The same launch, repeated with the caches disturbed in between, must give the same bits (tools/conv_repeat.py is the long
    form).  This is the screen for the failure class DESIGN 3.7 describes: a load still in flight when its destination register is
    re-used is right whenever it lands early - i.e. in every quiet, isolated run - and wrong under memory load; the split-K shapes
    with two or three chunks per workgroup (short loops) showed it first."""
    from blobctrl_amd import _lib
    Cin, HW, M = C1 + C2, H * W, B * H * W
    x1 = h(g(1, B, HW, C1))
    x2 = h(g(4, B, HW, C2)) if C2 else None
    w, b = g(2, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(3, Cout).cuda()
    gamma, beta = (1.0 + 0.2 * g(5, Cin)).cuda(), (0.3 * g(6, Cin)).cuda()
    thrash = torch.zeros(48 << 20, dtype=torch.float32, device="cuda")

    def fn():
        kw = dict(A2=x2, C1=C1, lda2=C2) if C2 else {}
        return rec.gemm(A=x1, lda=C1, W=wmat(w, tile), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout), bias=b,
                        conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1), rows_per_batch=HW, tile_cfg=tcfg(tile),
                        a_act=_lib.ACT_SILU, want_gn=True,
                        a_gn=dict(x1=x1, C1=C1, x2=x2, C2=C2, B=B, HW=HW, G=32, eps=1e-5, gamma=gamma, beta=beta), **kw)
    seg = rec.begin("repeat")
    out = fn()
    stream = torch.cuda.current_stream().cuda_stream
    ref = None
    for r in range(12):
        if r % 2:
            thrash.add_(1.0)
        seg.run(stream)
        torch.cuda.synchronize()
        o = out.clone()
        assert torch.isfinite(o.float()).all()
        if ref is None:
            ref = o
        assert torch.equal(o, ref), f"repeat {r} differs: max abs {float((o.float() - ref.float()).abs().max()):.3g}"


@pytest.mark.parametrize("B,H,W,C1,C2,Cout,sk", [(2, 64, 128, 320, 0, 320, 1), (1, 32, 64, 640, 320, 640, 2), (2, 8, 16, 1280, 1280, 1280, 5), (1, 8, 16, 320, 0, 160, 1)])
def test_wreg_conv_reads_the_time_embedding_row_of_the_current_step(rec, B, H, W, C1, C2, Cout, sk):
    """The ResBlock convolution as the ENGINE launches it: the time-embedding row comes from a per-edit table [steps][B][Cout] behind a
    device-side step counter (BcGemm.rowvec_idx / rowvec_step), GroupNorm finalized in the prologue from the statistics totals, residual,
    split-K.  Round 5 found the step counter's scalar load unprotected against the compiler moving its destination register (harmless in
    conv_wreg.hip by luck of the allocation, a memory fault in a sibling kernel): this is the kernel-level test the path did not have."""
    from blobctrl_amd import _lib
    Cin = C1 + C2
    x1 = g(1, B, C1, H, W) * 1.3 + 0.1
    x2 = g(2, B, C2, H, W) * 0.7 if C2 else None
    w, b = g(3, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), g(4, Cout)
    gamma, beta = 1.0 + 0.2 * g(5, Cin), 0.3 * g(6, Cin)
    temb, R = g(7, 3, B, Cout), g(8, B, Cout, H, W)
    M, HW, G = B * H * W, H * W, (32 if Cin % 32 == 0 else 8)
    for step in (2, 0):
        step_idx = torch.tensor([step], dtype=torch.int32, device="cuda:0")

        def fn():
            t1, t2 = nhwc(x1), (nhwc(x2) if C2 else None)
            kw = dict(A2=t2, C1=C1, lda2=C2) if C2 else {}
            return rec.gemm(A=t1, lda=C1, W=wmat(w, "wreg"), M=M, N=Cout, K=9 * Cin, out=rec.empty(M, Cout), bias=b.cuda(),
                            conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1), rows_per_batch=HW, tile_cfg=_lib.TILE_WREG, splitk=sk,
                            a_gn=dict(x1=t1, C1=C1, x2=t2, C2=C2, B=B, HW=HW, G=G, eps=1e-5, gamma=gamma.cuda(), beta=beta.cuda()), a_act=_lib.ACT_SILU,
                            rowvec=h(temb), ld_rowvec=Cout, rowvec_idx=step_idx, rowvec_step=B * Cout, R=nhwc(R), ldr=Cout, want_gn=True, **kw)
        out = run(rec, fn)
        xc = (torch.cat([x1, x2], 1) if C2 else x1).half().float()
        y = F.silu(F.group_norm(xc, G, gamma, beta, 1e-5))
        ref = F.conv2d(y.half().float(), w.half().float(), b, padding=1) + temb[step].half().float()[:, :, None, None] + R.half().float()
        close(from_nhwc(out, B, H, W), ref, rtol=4e-3, what=f"conv with the time-embedding row of step {step}")
