#!/usr/bin/env python3
"""Isolated timing of the transformer / 1x1 GEMM shapes of a 512^2 step (UNet at CFG batch 2) with the tile / split-K the step uses.
BC_GEMM_DBG ablation bits: 1 no MFMA, 2 no operand DMA in the loop, 4 no epilogue, 8 no fragment reads."""
import math
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
# (M, N, K, kind)
SHAPES = [(16384, 2560, 320, "geglu"), (16384, 320, 1280, "res"), (16384, 320, 320, "res"), (16384, 640, 320, "plain"),
          (4096, 5120, 640, "geglu"), (4096, 640, 2560, "res"), (4096, 640, 640, "res"),
          (1024, 10240, 1280, "geglu"), (1024, 1280, 5120, "res"), (1024, 1280, 1280, "res"), (8192, 320, 320, "res")]
for (M, N, K, kind) in SHAPES:
    A = torch.randn(M, K, device=dev, dtype=torch.float16)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half()
    bias = torch.randn(N, device=dev)
    n_out = N // 2 if kind == "geglu" else N
    out = torch.empty(M, n_out, device=dev, dtype=torch.float16)
    R = torch.randn(M, n_out, device=dev, dtype=torch.float16)
    seg = rec.begin("t")
    kw = dict(act=_lib.ACT_GEGLU) if kind == "geglu" else (dict(R=R, ldr=n_out) if kind == "res" else {})
    rec.gemm(A=A, W=W, M=M, N=N, K=K, out=out, bias=bias, **kw)
    us = sorted(time_launch(rec, seg, stream, 10) for _ in range(3))[1]
    print(f"{M}x{N}x{K} {kind:6s} {seg.meta[-1]['variant']:45s} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
