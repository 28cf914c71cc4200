"""Stand-alone GroupNorm(+SiLU) timing under hipGraph replay (no host launch overhead): a GEMM produces x and its per-channel
partials once, then a chain of N fused GroupNorm launches reads them."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blobctrl_amd.launch import Recorder            # noqa: E402


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    s = st.cuda_stream
    for (B, HW, C) in [(2, 8192, 320), (1, 8192, 320), (2, 2048, 640), (2, 512, 1280), (2, 128, 1280), (2, 8192, 640)]:
        rec = Recorder(dev)
        M = B * HW
        A = torch.randn(M, 64, device=dev, dtype=torch.float16)
        W = torch.randn(C, 64, device=dev, dtype=torch.float16) * 0.1
        gamma = torch.ones(C, device=dev)
        beta = torch.zeros(C, device=dev)
        seg0 = rec.begin("prod")
        x = rec.gemm(A=A, W=W, M=M, N=C, K=64, out=rec.empty(M, C), rows_per_batch=HW, want_gn=True)
        seg0.run(s)
        N = 50
        seg = rec.begin("gn")
        outs = [rec.empty(B, HW, C) for _ in range(2)]
        for i in range(N):
            rec.groupnorm(x, C, None, 0, B, HW, 32, 1e-5, gamma, beta, True, out=outs[i % 2])
        seg.run(s)
        torch.cuda.synchronize()
        seg.capture(s)
        for _ in range(3):
            seg.run(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            seg.run(s)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 10 / N * 1e6
        nbytes = 2 * B * HW * C * 2
        print(f"GN [{B},{HW},{C}]: {us:6.2f} us per launch (graph chain)  {nbytes / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
