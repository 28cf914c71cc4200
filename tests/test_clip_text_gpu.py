"""CLIP text encoder on the HIP path vs the transformers-generated fixture (tiny) and the CPU oracle
(full-size SD-1.5 text tower), plus the causal attention entry point on its own.  fp16 storage / fp32 accumulate: max-abs <= 1e-2
of the tensor scale, PSNR >= 40 dB."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from blobctrl_amd import _lib, synth  # noqa: E402
from tests.common import g, psnr  # noqa: E402


def _close(a, b, tol=1e-2, db=40.0):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err < tol and psnr(a, b) > db, f"max-abs/scale {err:.3e}, PSNR {psnr(a, b):.1f} dB"
    return err


@pytest.mark.parametrize("B,H,d,N", [(2, 4, 16, 77), (1, 12, 64, 77), (2, 2, 40, 200), (1, 2, 8, 64)])
def test_causal_attention_matches_torch(B, H, d, N):
    lib = _lib.load()
    Cc = H * d
    q, k, v = (g(s, B, N, Cc).half().cuda() for s in (1, 2, 3))
    ldvt = (N + 63) // 64 * 64
    vt = torch.zeros(B, Cc, ldvt, device="cuda", dtype=torch.float16)
    vt[:, :, :N] = v.transpose(1, 2)
    o = torch.empty(B, N, Cc, device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.bc_attention_causal(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, d, N, N, Cc, Cc, ldvt, Cc,
                                       N * Cc, N * Cc, Cc * ldvt, N * Cc, d ** -0.5, s), "bc_attention_causal")
    qh, kh, vh = (t.view(B, N, H, d).transpose(1, 2).float() for t in (q, k, v))
    mask = torch.full((N, N), float("-inf"), device="cuda").triu(1)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5 + mask, -1) @ vh).transpose(1, 2).reshape(B, N, Cc)
    assert (o.float() - ref).abs().max().item() < 2e-3
    # the entry point insists on a square problem
    assert lib.bc_attention_causal(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, d, N, N - 1, Cc, Cc, ldvt, Cc,
                                   N * Cc, N * Cc, Cc * ldvt, N * Cc, d ** -0.5, s) != 0


def test_clip_text_tiny_matches_transformers_fixture(golden_dir):
    import os
    from blobctrl_amd.clip_text import CLIPTextModel
    z = np.load(os.path.join(golden_dir, "clip_text_tiny.npz"))
    sd = synth.synth_state_dict(synth.clip_text_param_shapes(99, 64, 3, 128, 77), 77)
    enc = CLIPTextModel(sd, num_heads=4)
    out = enc(torch.from_numpy(z["ids"]))[0]
    assert out.shape == (2, 77, 64) and out.dtype == torch.float16
    _close(out.float().cpu().numpy(), z["last_hidden_state"])
    _close(enc(torch.from_numpy(z["ids"]), clip_skip=1)[0].float().cpu().numpy(), z["clip_skip_1"])
    _close(enc(torch.from_numpy(z["short_ids"]))[0].float().cpu().numpy(), z["short_last_hidden_state"])
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 78, dtype=torch.int64))
    with pytest.raises(NotImplementedError):
        enc(torch.from_numpy(z["ids"]), attention_mask=torch.ones(2, 77))
    # ids outside the vocabulary are clamped by the gather kernel (no out-of-bounds read)
    bad = torch.from_numpy(z["ids"]).clone()
    bad[0, 3] = 10 ** 6
    ok = torch.from_numpy(z["ids"]).clone()
    ok[0, 3] = 98
    assert torch.equal(enc(bad)[0], enc(ok)[0])


def test_clip_text_full_size_vs_oracle():
    from blobctrl_amd.clip_text import CLIPTextModel
    from oracle.clip_text import clip_text_hidden
    sd = synth.synth_state_dict(synth.clip_text_param_shapes(), 88)
    ids = torch.from_numpy(np.random.Generator(np.random.PCG64(9)).integers(0, 49408, size=(2, 77)).astype(np.int64))
    ref = clip_text_hidden(sd, ids, 12).numpy()
    out = CLIPTextModel(sd, num_heads=12)(ids)[0].float().cpu().numpy()
    e = _close(out, ref)
    print(f"full-size CLIP text encoder: max-abs/scale {e:.3e}, PSNR {psnr(out, ref):.1f} dB")


def test_pipeline_encode_prompt_feeds_the_loop(golden_dir):
    import os
    from blobctrl_amd.clip_text import CLIPTextModel
    from oracle.clip_text import clip_text_hidden
    from tests.common import TINY, tiny_weights
    from tests.gpu_common import make_pipeline
    c = TINY
    sd = synth.synth_state_dict(synth.clip_text_param_shapes(99, c["ctx"], 2, 32, 77), 31)      # hidden = the tiny cross-attn dim
    usd, bsd = tiny_weights()
    pipe = make_pipeline(usd, bsd, scheduler="ddim")
    with pytest.raises(ValueError):
        pipe.encode_prompt(torch.zeros(1, 77, dtype=torch.int64), torch.zeros(1, 77, dtype=torch.int64))
    pipe.text_encoder = CLIPTextModel(sd, num_heads=2)
    rng = np.random.Generator(np.random.PCG64(3))
    pos = torch.from_numpy(rng.integers(0, 99, size=(1, 77)).astype(np.int64))
    neg = torch.from_numpy(rng.integers(0, 99, size=(1, 77)).astype(np.int64))
    emb = pipe.encode_prompt(pos, neg)
    assert emb.shape == (2, 77, c["ctx"])
    ref = clip_text_hidden(sd, torch.cat([neg, pos]), 2).numpy()
    _close(emb.float().cpu().numpy(), ref)
    lat = pipe(emb, g(33, 1, 4, 8, 8), g(34, 1, 4, 8, 8), g(74, 1, 2, 8, 8).abs().clamp(max=1), g(75, 1, 1, c["feat"]),
               num_inference_steps=2, latents=g(76, 1, 4, 8, 8))
    assert lat.shape == (1, 4, 8, 8) and bool(torch.isfinite(lat).all())
