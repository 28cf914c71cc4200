"""The fused row-chain launches (csrc/rowchain.hip) against the unfused launch list of the same block (both through the C ABI;
the reference fixture for the block is tests/test_blocks_gpu.py::test_transformer_block): UNet form with the BlobNet residual added
on the right-hand part of the canvas and GroupNorm partials of the output, BlobNet form with the zero-conv and its device-side
conditioning scale, at several canvas shapes incl. 64-row blocks that straddle the r2 boundary."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.common import g, psnr  # noqa: E402
from tests.test_blocks_gpu import _act, _plan, _run  # noqa: E402


def _block_sd(cross, seed, zero=False, C=320):
    from blobctrl_amd import synth
    from tests.common import block_param_shapes
    sh = block_param_shapes("transformer", dict(C=C, ctx=768 if cross else None))
    sd = synth.synth_state_dict(sh, seed)
    if zero:
        sd["zero.weight"], sd["zero.bias"] = g(seed + 1, C, C, 1, 1) * 0.05, g(seed + 2, C) * 0.1
    return sd


def _record(monkeypatch, fused, cross, B, H, W, with_r2, zero, C=320, split=1, T=7, midx=True, ffp=True):
    from tests.common import set_plan
    # (the engine takes the 640-channel form only from 64 row blocks upwards; feed-forward of the block end over `split` workgroups per row
    #  block; BlobNet's blocks keep OUT_FF + OUT_TAIL by default: engine.rowchain_ffp)
    set_plan(monkeypatch, rowchain_min_blocks_640=1, rowchain_min_blocks_640_blob_up=1, ff_split_640=split, ff_split_640_blob=split, ff_split_320=split,
             midx=midx, rowchain=fused, ffp=ffp, ffp_blob=1)
    sd = _block_sd(cross, 77, zero, C)
    rec, seg, plan = _plan("rc", sd, B, H, W, heads=8, cross=768 if cross else None)
    x = g(5, B, C, H, W) * 1.3 + 0.2
    if cross:
        plan.record_context(g(6, B, T, 768).reshape(-1, 768).half().cuda(), T)
    r2 = None
    if with_r2:
        r2 = (g(7, 1, H * W, C) * 0.7).half().cuda()
        plan.res_bmod = 1
    zspec = None
    if zero:
        scale = torch.tensor([0.0, 0.8, 1.0], device="cuda:0")
        idx = torch.tensor([1], dtype=torch.int32, device="cuda:0")
        zspec = ("blk.zero" if False else "blk.zero", 1.0, scale, idx, 0)
    out, pre = plan.transformer("blk.", _act(x), r2=r2, zero=zspec)
    if zero and pre is None:            # the unfused path leaves the zero-conv to the caller (engine._Feats.append)
        from blobctrl_amd import _lib
        M = B * H * W
        pre = plan.dense(out.t, M, C, "blk.zero", C, kind="zero_conv", alpha=1.0, alpha_dev=zspec[2], alpha_idx=zspec[3],
                         alpha_bstride=0, rows_per_batch=H * W)
    part = rec.tots.get(out.t.data_ptr())
    _run(seg)
    return rec, out, pre, part


@pytest.mark.parametrize("cross,B,H,W,with_r2,zero,C,split", [
    (True, 2, 16, 32, True, False, 320, 1), (True, 1, 8, 24, True, False, 320, 1), (False, 1, 16, 32, False, True, 320, 1),
    (False, 2, 8, 8, False, True, 320, 1), (True, 2, 8, 8, True, False, 320, 1), (True, 2, 16, 32, True, False, 640, 1),
    (False, 1, 8, 24, False, True, 640, 1), (True, 1, 8, 8, True, False, 640, 1),
    # the block end as OUT_FF + OUT_TAIL, the hidden chunks over 2 / 4 / 5 workgroups per row block
    (True, 2, 16, 32, True, False, 640, 2), (False, 1, 8, 24, False, True, 640, 4), (True, 1, 8, 8, True, False, 640, 5),
    (True, 2, 16, 32, True, False, 320, 2), (False, 1, 16, 32, False, True, 320, 5)])
def test_rowchain_matches_the_unfused_block(monkeypatch, cross, B, H, W, with_r2, zero, C, split):
    rec_f, out_f, pre_f, part_f = _record(monkeypatch, True, cross, B, H, W, with_r2, zero, C, split)
    # split block end (round 5): OUT_FFP (every slice through proj_out [+ zero-conv]) + the sum of the fp16 partial outputs
    assert any("out_ffp/" in (m["variant"] or "") for m in rec_f.seg.meta) == (split > 1)
    assert any(m["kind"] == "rowchain_sum" for m in rec_f.seg.meta) == (split > 1)
    _, out_u, pre_u, _ = _record(monkeypatch, False, cross, B, H, W, with_r2, zero, C)
    if split > 1:        # the round-3 form (OUT_FF + OUT_TAIL over fp32 partial sums) stays available behind BC_PLAN ffp=0
        rec_t, out_t, pre_t, _ = _record(monkeypatch, True, cross, B, H, W, with_r2, zero, C, split, ffp=False)
        assert any("out_tail" in (m["variant"] or "") for m in rec_t.seg.meta) and not any(m["kind"] == "rowchain_sum" for m in rec_t.seg.meta)
        t_, u_ = out_t.t.float().cpu().numpy(), out_u.t.float().cpu().numpy()
        assert np.abs(t_ - u_).max() / np.abs(u_).max() < 6e-3 and psnr(t_, u_) > 50.0
        if zero:
            zt, zu = pre_t.float().cpu().numpy().reshape(-1), pre_u.float().cpu().numpy().reshape(-1)
            assert np.abs(zt - zu).max() / np.abs(zu).max() < 6e-3
    a, b = out_f.t.float().cpu().numpy(), out_u.t.float().cpu().numpy()
    rel = np.abs(a - b).max() / np.abs(b).max()
    print(f"row-chain vs unfused block output: max-abs/scale {rel:.3e}, PSNR {psnr(a, b):.1f} dB")
    assert rel < 6e-3 and psnr(a, b) > 50.0
    if zero:
        za, zb = pre_f.float().cpu().numpy().reshape(-1), pre_u.float().cpu().numpy().reshape(-1)
        relz = np.abs(za - zb).max() / np.abs(zb).max()
        print(f"   zero-conv residual: max-abs/scale {relz:.3e}")
        assert relz < 6e-3
    # GroupNorm statistics of the fused output: the totals equal the per-(image, channel) sums of the fp16 tensor
    from blobctrl_amd.launch import decode_gn_tot, gn_tot_slots
    s = gn_tot_slots(decode_gn_tot(part_f).float())
    o = out_f.t.float().cpu().view(B, H * W, C)
    want = gn_tot_slots(torch.stack([o.sum(1), (o * o).sum(1)], -1))
    assert torch.allclose(s[..., 0], want[..., 0], rtol=1e-3, atol=1e-2 * (H * W) ** 0.5)
    assert torch.allclose(s[..., 1], want[..., 1], rtol=1e-3, atol=1e-2 * (H * W) ** 0.5)


@pytest.mark.parametrize("B,H,W,C,T", [(2, 16, 32, 320, 77), (1, 8, 24, 320, 80), (2, 8, 8, 320, 1), (2, 16, 32, 640, 77), (1, 8, 8, 640, 48),
                                       (1, 8, 24, 640, 17)])
def test_cross_attention_inside_the_mid_launch(monkeypatch, B, H, W, C, T):
    """BC_CHAIN_MIDX (to_out + residual -> LayerNorm2 -> to_q -> softmax(q K^T) V over the context tokens, one launch) against the same
    block with BC_CHAIN_MID + bc_attention, and against the unfused launch list: 77 tokens (the CLIP context), the 80-token maximum, a
    single token (softmax = 1), and counts that leave key tiles partly or wholly masked."""
    rec_x, out_x, _, _ = _record(monkeypatch, True, True, B, H, W, True, False, C, 1, T=T, midx=True)
    assert any("midx" in (m["variant"] or "") for m in rec_x.seg.meta)
    assert not any(m["kind"] == "attention" and m["shape"][-1] == T for m in rec_x.seg.meta)       # the cross-attention launch is gone
    rec_m, out_m, _, _ = _record(monkeypatch, True, True, B, H, W, True, False, C, 1, T=T, midx=False)
    assert not any("midx" in (m["variant"] or "") for m in rec_m.seg.meta)
    _, out_u, _, _ = _record(monkeypatch, False, True, B, H, W, True, False, C, 1, T=T)
    a, b, c = (o.t.float().cpu().numpy() for o in (out_x, out_m, out_u))
    rel_m, rel_u = np.abs(a - b).max() / np.abs(b).max(), np.abs(a - c).max() / np.abs(c).max()
    print(f"MIDX vs MID + attention: {rel_m:.3e} ({psnr(a, b):.1f} dB); vs the unfused block: {rel_u:.3e} ({psnr(a, c):.1f} dB)")
    assert rel_m < 4e-3 and psnr(a, b) > 54.0
    assert rel_u < 6e-3 and psnr(a, c) > 50.0


@pytest.mark.parametrize("C,T,B", [(320, 77, 2), (640, 77, 2), (320, 80, 1), (640, 5, 1)])
def test_kv_stream_kernel_equals_the_host_packing(C, T, B):
    """bc_rowchain_pack_kv (device) and weights.pack_rowchain_kv (host statement of the layout) write the same fragment streams."""
    from blobctrl_amd import _lib
    from blobctrl_amd.weights import pack_rowchain_kv
    lib = _lib.load()
    ldvt = 128
    k = (g(31, B, T, C) * 2).half().cuda()
    vt = (g(32, B, C, ldvt) * 2).half().cuda()
    frags = lib.bc_rowchain_kv_frags(C)
    out = torch.full((B * (C // 80) * frags * 64 * 8,), 7.0, dtype=torch.float16, device="cuda:0")
    _lib.check(lib.bc_rowchain_pack_kv(k.data_ptr(), C, vt.data_ptr(), ldvt, B, T, C, out.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "bc_rowchain_pack_kv")
    torch.cuda.synchronize()
    ref = pack_rowchain_kv(k, vt, T)
    assert ref.numel() == out.numel() and torch.equal(out.view_as(ref).cpu(), ref.cpu())
    with pytest.raises(_lib.BlobCtrlHipError):
        _lib.check(lib.bc_rowchain_pack_kv(k.data_ptr(), C, vt.data_ptr(), ldvt, B, 81, C, out.data_ptr(), 0), "bc_rowchain_pack_kv")


def test_rowchain_rejects_unsupported_shapes():
    from blobctrl_amd import _lib
    lib = _lib.load()
    assert lib.bc_rowchain_supported(320, 2 * 8192, 8192) == 1 and lib.bc_rowchain_supported(640, 4096, 2048) == 1
    assert lib.bc_rowchain_supported(1280, 1024, 512) == 0 and lib.bc_rowchain_supported(320, 96, 96) == 0
    assert lib.bc_rowchain_stream_frags(320, 0, 0, 1) == 220 and lib.bc_rowchain_stream_frags(320, 2, 1, 1) == 770
    assert lib.bc_rowchain_stream_frags(640, 1, 0, 1) == 220 and lib.bc_rowchain_stream_frags(640, 2, 0, 1) == 1420
    # split block end: OUT_FF = to_out + 20 / nsplit chunks of 60 fragments (+ 20 padding), OUT_TAIL = proj_out [+ zero-conv]
    assert lib.bc_rowchain_stream_frags(640, 3, 0, 2) == 100 + 10 * 60 + 20 and lib.bc_rowchain_stream_frags(640, 3, 0, 3) == -1
    assert lib.bc_rowchain_kv_frags(320) == 2 * (10 + 9) + 20 and lib.bc_rowchain_kv_frags(640) == 15 + 15 + 20 and lib.bc_rowchain_kv_frags(1280) == -1
    assert lib.bc_rowchain_stream_frags(320, 5, 0, 1) == lib.bc_rowchain_stream_frags(320, 1, 0, 1)       # MIDX reads MID's stream
    assert lib.bc_rowchain_stream_frags(640, 4, 1, 2) == 220 and lib.bc_rowchain_stream_frags(320, 3, 0, 5) == 50 + 2 * 60 + 20
