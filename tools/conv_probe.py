#!/usr/bin/env python3
"""A/B probe of the ResBlock convolution path on the production shapes, one process, interleaved:
  old = GroupNorm-apply pass (gn_apply_fused) + implicit-GEMM kernel (tuned tile / split-K)
  new = gn_finalize (per-channel affine) + conv_halo.hip with the normalise + SiLU fused into the halo staging
Reports microseconds per (GroupNorm + conv) pair and TFLOP/s of the convolution's 2*M*N*K."""
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder, encode_gn_tot  # noqa: E402
from blobctrl_amd.weights import pack_conv_wreg  # noqa: E402
from tools.tune_gemm import time_launch, time_launch_cold  # noqa: E402

dev = torch.device("cuda:0")
rec = Recorder(dev)
stream = torch.cuda.current_stream().cuda_stream
# (B, H, W, C1, C2, Cout): UNet (B = 2) ResBlock convolutions of a 512^2 edit; BlobNet runs the same at B = 1
SHAPES = [(2, 64, 128, 320, 0, 320), (2, 64, 128, 640, 320, 320), (2, 64, 128, 320, 320, 320), (1, 64, 128, 320, 0, 320),
          (2, 32, 64, 640, 0, 640), (2, 32, 64, 320, 0, 640), (2, 32, 64, 1280, 640, 640), (1, 32, 64, 640, 0, 640),
          (2, 16, 32, 1280, 0, 1280), (2, 16, 32, 1280, 1280, 1280), (1, 16, 32, 1280, 0, 1280),
          (2, 8, 16, 1280, 0, 1280), (2, 8, 16, 1280, 1280, 1280), (1, 8, 16, 1280, 0, 1280),
          # (round 6) the UNet's shapes at batch 8: eight rounds of workgroups per launch
          (16, 64, 128, 320, 0, 320), (16, 32, 64, 640, 0, 640), (16, 16, 32, 1280, 0, 1280)]
only = os.environ.get("PROBE_ONLY")
cold = os.environ.get("PROBE_COLD")            # weights from HBM (caches flushed before every launch), activations warm: as in the step
thrash = torch.zeros(160 << 20, dtype=torch.float32, device=dev) if cold else None
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["PROBE_SHAPES"].split(",")]
for (B, H, W, C1, C2, Co) in SHAPES:
    Cin, HW, M = C1 + C2, H * W, B * H * W
    x1 = torch.randn(B, HW, C1, device=dev, dtype=torch.float16)
    x2 = torch.randn(B, HW, C2, device=dev, dtype=torch.float16) if C2 else None
    wt = (torch.randn(Co, 9 * Cin, device=dev) / math.sqrt(9 * Cin)).half()
    wt_wreg = pack_conv_wreg(wt)
    bias = torch.randn(Co, device=dev)
    gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1)
    flops = 2.0 * M * Co * 9 * Cin
    # per-channel partial statistics as the producer GEMMs would have left them (neither path then runs a statistics pass)
    for t, c in ((x1, C1), (x2, C2)):
        if t is not None:
            ns = HW // 128
            f = t.float().view(B, ns, 128, c)
            rec.tots[t.data_ptr()] = encode_gn_tot(torch.stack([f.sum((1, 2)), (f * f).sum((1, 2))], -1)).to(dev)
    res = {}
    seg = rec.begin("old")
    y = rec.groupnorm(x1, C1, x2, C2, B, HW, 32, 1e-5, gamma, beta, True)
    rec.gemm(A=y, W=wt, M=M, N=Co, K=9 * Cin, out=rec.empty(M, Co), bias=bias, conv=conv, rows_per_batch=HW, want_gn=True)
    old_variant = seg.meta[-1]["variant"] + str(seg.meta[-1]["shape"][-1])
    segs = {"old": seg}
    for sk in ([None] if not only else [None if v == "auto" else int(v) for v in only.split(",")]) + ([1, 2, 4, 5, 10] if not only else []):
        nch = Cin // 64
        if sk is not None and (sk > nch or -(-nch // sk) > 40 or (B * H * W // 128) * (Co // 160) * sk > 1024):
            continue
        kw = dict(A2=x2, C1=C1, lda2=C2) if C2 else {}
        for name, cfg, wm in (("halo", _lib.TILE_HALO, wt), ("wreg", _lib.TILE_WREG, wt_wreg)):
            s2 = rec.begin(f"{name}_sk{sk}")
            rec.gemm(A=x1, lda=C1, W=wm, M=M, N=Co, K=9 * Cin, out=rec.empty(M, Co), bias=bias, conv=conv, rows_per_batch=HW,
                     tile_cfg=cfg, splitk=sk, a_act=_lib.ACT_SILU, want_gn=not os.environ.get("PROBE_NO_GN"),
                     a_gn=dict(x1=x1, C1=C1, x2=x2, C2=C2, B=B, HW=HW, G=32, eps=1e-5, gamma=gamma, beta=beta), **kw)
            segs[f"{name}_sk{sk}({s2.meta[-1]['shape'][-1]})"] = s2
    # (round 6) GroupNorm-apply pass + the PLAIN conv_wreg kernel (no statistics table, no in-place transform of the halo rows)
    s3 = rec.begin("gnpass_wreg")
    y3 = rec.groupnorm(x1, C1, x2, C2, B, HW, 32, 1e-5, gamma, beta, True)
    rec.gemm(A=y3, W=wt_wreg, M=M, N=Co, K=9 * Cin, out=rec.empty(M, Co), bias=bias, conv=conv, rows_per_batch=HW, tile_cfg=_lib.TILE_WREG,
             want_gn=not os.environ.get("PROBE_NO_GN"))
    segs[f"gnpass+wreg0({s3.meta[-1]['shape'][-1]})"] = s3
    for rnd in range(3):
        for k, s in segs.items():
            res.setdefault(k, []).append(time_launch_cold(rec, s, stream, 6, thrash, [t for t in (x1, x2) if t is not None])
                                         if cold else time_launch(rec, s, stream, 10))
    line = f"B{B} {H}x{W} {C1}+{C2}->{Co}: old[{old_variant}]"
    for k, v in res.items():
        us = sorted(v)[1]
        line += f" | {k} {us:7.1f} us {flops / us / 1e6:6.0f} TF"
    print(line, flush=True)
    if os.environ.get("PROBE_SPLIT"):           # per-kernel HIP-event times of the halo segments (finalize | conv main | split-K reducer)
        for k, sgm in segs.items():
            best = None
            for _ in range(4):
                if cold:
                    thrash.add_(1)
                    for t in (x1, x2):
                        if t is not None:
                            t.mul_(1)
                rows = sgm.run_timed_kernels(stream)
                cur = [(m["variant"] or m["kind"], round(a * 1e3, 1), round(r * 1e3, 1)) for m, a, r in rows]
                best = cur if best is None or sum(c[1] + c[2] for c in cur) < sum(c[1] + c[2] for c in best) else best
            print("      ", k, " | ".join(f"{n.split('<')[0]} {a}" + (f" + reduce {r}" if r else "") for n, a, r in best), flush=True)
