"""What the vendor library reaches on this box for the same problems as tools/gemm256_probe.hip (calibration only: torch.matmul ->
hipBLASLt / rocBLAS, fp16 in, fp32 accumulate, random-normal operands - power-limited clocks make zeros / constants meaningless)."""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 2560, 320), (16384, 5120, 640), (4096, 5120, 640),
                  # (round 6) the shapes csrc/gemm256.hip runs in the batch-8 plan (tools/gemm8p_probe.hip on the same box)
                  (8192, 10240, 1280), (8192, 1280, 5120), (8192, 1280, 1280), (8192, 3840, 1280), (4096, 10240, 1280), (9216, 10240, 1280), (8192, 1280, 6400)]:
    A = torch.randn(M, K, device=dev, dtype=torch.float16) * 1.0
    B = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    for _ in range(5):
        C = A @ B.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C = A @ B.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"torch.matmul fp16 {M} x {N} x {K}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
