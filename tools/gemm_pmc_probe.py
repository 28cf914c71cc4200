#!/usr/bin/env python3
"""One launch per tile configuration of one conv-shaped and one dense GEMM, for rocprofv3 --pmc runs."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from blobctrl_amd.launch import Recorder  # noqa: E402
os.environ["BC_NO_TUNING"] = "1"
dev = torch.device("cuda:0")
rec = Recorder(dev)
stream = torch.cuda.current_stream().cuda_stream
M, N, K = 4096, 4096, 4096
A = torch.randn(M, K, device=dev, dtype=torch.float16)
W = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
out = torch.empty(M, N, device=dev, dtype=torch.float16)
for cfg in (1, 3):
    seg = rec.begin("t")
    rec.gemm(A=A, W=W, M=M, N=N, K=K, out=out, tile_cfg=cfg, splitk=1)
    for _ in range(3):
        seg.run(stream)
torch.cuda.synchronize()
