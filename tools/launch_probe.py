"""Per-node cost of a hipGraph chain of dependent tiny kernels (how much of a step is launch / dependency overhead)."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blobctrl_amd.launch import Recorder            # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rec = Recorder(dev)
    st = torch.cuda.Stream()
    s = st.cuda_stream
    for n_elem in (64, 1 << 20):
        x = torch.randn(n_elem, device=dev, dtype=torch.float16)
        y = torch.empty_like(x)
        for N in (100, 400):
            seg = rec.begin("chain")
            for i in range(N):
                a, b = (x, y) if i % 2 == 0 else (y, x)
                rec.call("bc_silu", a.data_ptr(), b.data_ptr(), n_elem, kind="silu", keep=(a, b))
            for graph in (False, True):
                if graph:
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
                dt = (time.perf_counter() - t0) / 10
                print(f"n_elem={n_elem:8d} chain of {N:4d} kernels, graph={graph}: {dt * 1e3:8.3f} ms  -> {dt / N * 1e6:6.2f} us/kernel",
                      flush=True)
            seg.release()


if __name__ == "__main__":
    main()
