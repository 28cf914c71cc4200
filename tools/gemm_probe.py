#!/usr/bin/env python3
"""Stand-alone GEMM probe: time every tile configuration on a few large dense shapes (kernel ceiling, no tail effects)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder  # noqa: E402
from tools.tune_gemm import time_launch  # noqa: E402

os.environ["BC_NO_TUNING"] = "1"
dev = torch.device("cuda:0")
rec = Recorder(dev)
stream = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(8192, 8192, 8192), (4096, 4096, 4096), (16384, 1280, 2880), (16384, 320, 2880)]:
    A = torch.randn(M, K, device=dev, dtype=torch.float16)
    W = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for cfg in range(1, 8):
        seg = rec.begin("t")
        rec.gemm(A=A, W=W, M=M, N=N, K=K, out=out, tile_cfg=cfg, splitk=1)
        us = time_launch(rec, seg, stream, 5)
        print(f"{M}x{N}x{K} {_lib.TILE_NAMES[cfg]:12s} {us:9.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
