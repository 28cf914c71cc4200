#!/usr/bin/env python3
"""(GPU) csrc/gemm256.hip as shipped (persistent workgroups, slab epilogue, bias) through the C ABI on random operands: TFLOP/s per shape,
back-to-back launches on rotating output buffers.  Companion of tools/gemm8p_probe.hip (the main loop alone with a plain store epilogue)
and tools/blas_ceiling.py (the vendor library on the same box)."""
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder  # noqa: E402

dev = torch.device("cuda:0")
rec = Recorder(dev)
s = torch.cuda.current_stream().cuda_stream
SHAPES = [(8192, 10240, 128), (8192, 10240, 256), (8192, 10240, 640), (8192, 10240, 2560), (8192, 10240, 5120),      # (K sweep: per-tile overhead vs per-k-tile time)
          (4096, 4096, 4096), (8192, 8192, 8192), (8192, 10240, 1280), (8192, 1280, 5120), (8192, 1280, 1280), (8192, 3840, 1280), (8192, 1280, 6400),
          (4096, 10240, 1280), (9216, 10240, 1280), (16384, 5120, 640), (16384, 2560, 640)]
if os.environ.get("G256_SHAPES"):                        # e.g. G256_SHAPES=8192x10240x128,8192x10240x1280
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["G256_SHAPES"].split(",")]
for (M, N, K) in SHAPES:
    A = (torch.randn(M, K, device=dev) * 1.0).half()
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).half()
    b = torch.randn(N, device=dev)
    for act, name in ((_lib.ACT_NONE, "bias"), (_lib.ACT_GEGLU, "geglu")):
        if act == _lib.ACT_GEGLU and N != 10240:
            continue
        n_out = N // 2 if act == _lib.ACT_GEGLU else N
        seg = rec.begin(f"g{M}_{N}_{K}_{name}")
        outs = [rec.empty(M, n_out) for _ in range(4)]
        for o in outs:
            rec.gemm(A=A, W=W, M=M, N=N, K=K, out=o, bias=b, act=act, tile_cfg=_lib.TILE_G256)
        for _ in range(2):
            seg.run(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            seg.run(s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"gemm256 (library, {name:5s}) {M:6d} x {N:6d} x {K:5d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
