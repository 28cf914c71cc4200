"""Time the full-size SD-1.5 VAE (encode 512^2 / decode 64x64 latent) on the HIP path and print the per-kernel event table."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blobctrl_amd import synth                       # noqa: E402
from blobctrl_amd.vae import AutoencoderKL           # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # images per call (scripts/blobctrl_inference.py decodes num_samples = 2 at once)
    sd = synth.synth_state_dict(synth.vae_param_shapes(), 33)
    vae = AutoencoderKL(sd)
    z = torch.randn(B, 4, res // 8, res // 8).cuda()
    x = torch.randn(1, 3, res, res).clamp(-1, 1).cuda()
    for name, fn in (("decode", lambda: vae.decode(z)), ("encode", lambda: vae.encode(x))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"{name} {res}x{res} batch {z.shape[0] if name == 'decode' else 1}: {a.elapsed_time(b) / 10:.3f} ms")
    for key, P in vae._plans.items():
        rows = P.seg.run_timed(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rows = P.seg.run_timed(torch.cuda.current_stream().cuda_stream)
        agg = {}
        for m, ms in rows:
            k = (m["kind"], m["variant"], str(m["shape"]))
            e = agg.setdefault(k, [0, 0.0, 0])
            e[0] += 1
            e[1] += ms
            e[2] += m["flops"]
        tot = sum(ms for _, ms in rows)
        print(f"--- {key}: {len(rows)} launches, {tot:.3f} ms serial, {P.seg.flops / 1e12:.3f} TFLOP")
        for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
            tf = e[2] / (e[1] * 1e-3) / 1e12 if e[1] > 0 else 0
            print(f"  {e[1]:8.3f} ms  x{e[0]:<3d} {tf:7.1f} TF/s  {k[0]:12s} {k[1]:50s} {k[2]}")


if __name__ == "__main__":
    main()
