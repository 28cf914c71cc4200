"""Time bc_attention at the denoise loop's shapes (TFLOP/s algorithmic = 4 B H Nq Nkv d / t) and check against torch SDPA."""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blobctrl_amd import _lib                       # noqa: E402


def run(lib, B, H, d, Nq, Nkv, iters=20, check=True):
    torch.manual_seed(0)
    Cc = H * d
    q = torch.randn(B, Nq, Cc, device="cuda", dtype=torch.float16)
    k = torch.randn(B, Nkv, Cc, device="cuda", dtype=torch.float16)
    v = torch.randn(B, Nkv, Cc, device="cuda", dtype=torch.float16)
    ldvt = (Nkv + 63) // 64 * 64
    vt = torch.zeros(B, Cc, ldvt, device="cuda", dtype=torch.float16)
    vt[:, :, :Nkv] = v.transpose(1, 2)
    o = torch.empty(B, Nq, Cc, device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream().cuda_stream

    def go():
        _lib.check(lib.bc_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, d, Nq, Nkv, Cc, Cc, ldvt, Cc,
                                    Nq * Cc, Nkv * Cc, Cc * ldvt, Nq * Cc, d ** -0.5, s), "bc_attention")
    go()
    torch.cuda.synchronize()
    err = float("nan")
    if check:
        qh, kh, vh = (t.view(B, -1, H, d).transpose(1, 2).float() for t in (q, k, v))
        ref = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh
        ref = ref.transpose(1, 2).reshape(B, Nq, Cc)
        err = (o.float() - ref).abs().max().item()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        go()
    a.record()
    for _ in range(iters):
        go()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    tf = 4.0 * B * H * Nq * Nkv * d / (ms * 1e-3) / 1e12
    print(f"B={B} H={H} d={d:3d} Nq={Nq:5d} Nkv={Nkv:5d}: {ms * 1e3:8.1f} us  {tf:7.1f} TF/s  max-abs err {err:.2e}", flush=True)


def main():
    lib = _lib.load()
    if len(sys.argv) > 1:                               # A/B: another build of the library (same C ABI)
        alt = C.CDLL(os.path.abspath(sys.argv[1]))
        alt.bc_attention.argtypes = lib.bc_attention.argtypes
        alt.bc_attention.restype = C.c_int
        alt.bc_last_error.restype = C.c_char_p
        lib = alt
    print("lib =", sys.argv[1] if len(sys.argv) > 1 else "default")
    if os.environ.get("ATT_ONLY40"):
        run(lib, 2, 8, 40, 8192, 8192, check=False)
        run(lib, 4, 8, 40, 9216 * 2, 9216 * 2, iters=5, check=False)
        return
    for B in (2, 1):
        run(lib, B, 8, 40, 8192, 8192)
        run(lib, B, 8, 80, 2048, 2048)
        run(lib, B, 8, 160, 512, 512)
        run(lib, B, 8, 160, 128, 128)
    run(lib, 2, 8, 40, 8192, 77)
    run(lib, 2, 8, 80, 2048, 77)
    run(lib, 1, 16, 64, 257, 257)
    run(lib, 4, 8, 40, 9216 * 2, 9216 * 2, iters=5, check=False)


if __name__ == "__main__":
    main()
