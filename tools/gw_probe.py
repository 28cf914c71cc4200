#!/usr/bin/env python3
"""gemm_wreg.hip against gemm_fast.hip on the projections of the 1280-channel levels: correctness vs fp32 torch and time per launch
(HIP events over a graph of 20 back-to-back launches on rotating weight copies = cold weights, and on one copy = hot)."""
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder  # noqa: E402
from blobctrl_amd.weights import pack_gemm_wreg, fold_layernorm  # noqa: E402

dev = torch.device("cuda:0")
NCOPY = 24


def timed(rec, build, reps=5):
    """build(i) records launch i of NCOPY; returns us per launch (graph replay)."""
    seg = rec.begin("t")
    for i in range(NCOPY):
        build(i)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        seg.run(s.cuda_stream)
        s.synchronize()
        seg.capture(s.cuda_stream)
        seg.run(s.cuda_stream)
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            seg.run(s.cuda_stream)
        e1.record(s)
        s.synchronize()
    seg.release()
    return e0.elapsed_time(e1) * 1e3 / reps / NCOPY


def main():
    rec = Recorder(dev)
    torch.manual_seed(0)
    shapes = [(1024, 1280, 1280, "none"), (1024, 3840, 1280, "qkv_ln"), (1024, 10240, 1280, "geglu_ln"), (1024, 1280, 5120, "res"),
              (512, 1280, 1280, "none"), (256, 1280, 1280, "none"), (256, 10240, 1280, "geglu_ln"), (256, 1280, 5120, "res"),
              (1024, 1280, 2560, "res"), (1024, 1280, 640, "none")]
    only = sys.argv[1:] and [int(x) for x in sys.argv[1].split(",")]
    for si, (M, N, K, mode) in enumerate(shapes):
        if only and si not in only:
            continue
        A = (torch.randn(M, K, device=dev) * 1.5 + 0.3).half()
        Ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).half() for _ in range(NCOPY)]
        bias = torch.randn(N, device=dev)
        gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        R = torch.randn(M, N, device=dev).half()
        ln = mode.endswith("_ln")
        act = _lib.ACT_GEGLU if mode.startswith("geglu") else _lib.ACT_NONE
        n_out = N // 2 if act == _lib.ACT_GEGLU else N
        # ---- reference (fp32 on the fp16-rounded inputs)
        x = A.float()
        if ln:
            x = torch.nn.functional.layer_norm(x, (K,), gamma, beta, 1e-5)
        ref = x @ Ws[0].float().t() + bias
        if act == _lib.ACT_GEGLU:
            r4 = ref.view(M, N // 64, 2, 32)
            ref = (r4[:, :, 0] * torch.nn.functional.gelu(r4[:, :, 1])).reshape(M, N // 2)
        if mode == "res":
            ref = ref + R.float()
        # ---- baseline: what the engine runs today (LayerNorm launch + gemm_fast with the tuned tile)
        outs = [rec.empty(M, n_out) for _ in range(NCOPY)]
        lnb = rec.empty(M, K)

        def base(i):
            a = A
            if ln:
                rec.layernorm(A, M, K, gamma, beta, 1e-5, out=lnb)
                a = lnb
            if mode == "qkv_ln":
                rec.gemm(A=a, W=Ws[i], M=M, N=2560, K=K, out=outs[i], ldc=N, bias=bias)
                rec.gemm(A=a, W=Ws[i], w_offset=2560 * K, M=M, N=1280, K=K, out=outs[i], out_offset=2560, ldc=N)
            else:
                rec.gemm(A=a, W=Ws[i], M=M, N=N, K=K, out=outs[i], bias=bias, act=act, R=R if mode == "res" else None, ldr=N)
        t_base = timed(rec, base)
        line = f"[{si}] M={M:5d} N={N:5d} K={K:5d} {mode:9s} base {t_base:6.1f} us"
        if os.environ.get("PROBE_HOT"):
            line += f" (hot {timed(rec, lambda i: base(0)):5.1f})"
        for cfg in (_lib.TILE_GW64x128, _lib.TILE_GW64x256, _lib.TILE_GW64x320):
            nt = _lib.GW_TILES[cfg]
            if N % (64 * nt) or (mode == "qkv_ln" and 2560 % (64 * nt)):
                continue
            packs = []
            for i in range(NCOPY):
                w, cs, b2 = Ws[i], None, bias
                if ln:
                    w, cs, b2 = fold_layernorm(w, bias, gamma, beta)
                packs.append((pack_gemm_wreg(w, nt), cs, b2))
            vts = [rec.zeros(1, 1280, M) for _ in range(NCOPY)] if mode == "qkv_ln" else None
            out2 = [rec.empty(M, 2560 if mode == "qkv_ln" else n_out) for _ in range(NCOPY)]

            def gw(i):
                w, cs, b2 = packs[i]
                kw = dict(C_t=vts[i], ldc_t=M, n_t0=2560, rows_per_batch=M) if mode == "qkv_ln" else {}
                rec.gemm(A=A, W=w, M=M, N=N, K=K, out=out2[i], bias=b2, act=act, R=R if mode == "res" else None, ldr=N, tile_cfg=cfg,
                         ln_colsum=cs, **kw)
            t = timed(rec, gw)
            if os.environ.get("PROBE_HOT"):
                line += f" (hot {timed(rec, lambda i: gw(0)):5.1f})"
            if mode == "qkv_ln":
                got = torch.cat([out2[0].float(), vts[0][0].float().t()], 1)
            else:
                got = out2[0].float()
            err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            line += f" | {_lib.TILE_NAMES[cfg]} {t:6.1f} us err {err:.1e}"
            if mode == "qkv_ln" and err > 1e-2:
                d = (got - ref).abs()
                line += f" [qk part {d[:, :2560].max().item():.2e}, v part {d[:, 2560:].max().item():.2e}, v rows bad {(d[:, 2560:].max(1).values > 0.1).sum().item()}, v cols bad {(d[:, 2560:].max(0).values > 0.1).sum().item()}]"
        print(line, flush=True)


if __name__ == "__main__":
    main()
