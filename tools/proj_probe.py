#!/usr/bin/env python3
"""Projection-shaped GEMMs of the transformer blocks: normal vs transposed (V^T) output, per tile configuration."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder  # noqa: E402
from tools.tune_gemm import time_launch  # noqa: E402

dev = torch.device("cuda:0")
rec = Recorder(dev)
stream = torch.cuda.current_stream().cuda_stream
for (M, N, K, B) in [(16384, 320, 320, 2), (16384, 640, 320, 2), (16384, 960, 320, 2), (8192, 320, 320, 1), (4096, 640, 640, 2),
                     (4096, 1920, 640, 2), (1024, 1280, 1280, 2), (1024, 3840, 1280, 2)]:
    A = torch.randn(M, K, device=dev, dtype=torch.float16)
    W = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    rpb = M // B
    outT = torch.zeros(B, N, rpb, device=dev, dtype=torch.float16)
    res = []
    for mode in ("normal", "T"):
        best = None
        for cfg in range(0, 8):
            for sk in ((None,) if cfg == 0 else (1, 2)):
                seg = rec.begin("t")
                try:
                    if mode == "normal":
                        rec.gemm(A=A, W=W, M=M, N=N, K=K, out=out, tile_cfg=cfg, splitk=sk)
                    else:
                        rec.gemm(A=A, W=W, M=M, N=N, K=K, out=outT, out_mode=_lib.OUT_F16_T, ldc=rpb, rows_per_batch=rpb,
                                 tile_cfg=cfg, splitk=sk)
                    us = time_launch(rec, seg, stream, 10)
                except Exception as e:      # noqa: BLE001
                    continue
                tag = f"{_lib.TILE_NAMES[cfg]}/sk{sk}"
                if cfg == 0:
                    res.append(f"{mode} auto {us:6.1f} us")
                if best is None or us < best[0]:
                    best = (us, tag)
        res.append(f"{mode} best {best[0]:6.1f} us ({best[1]})")
    print(f"{M}x{N}x{K}: " + "; ".join(res) + f"   [{2.0 * M * N * K / 1e6:.0f} MFLOP]", flush=True)
