"""Is the step clock / power limited?  The same 50-step edit (same launches, same shapes) on the seeded random weights and on
weights whose matrices are all zero (activations collapse to the biases: the kernels do the same work on data that toggles few bits).
Usage (GPU): python tools/power_probe.py"""
import os
import sys
import time
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import splat_features
    dev = torch.device("cuda:0")
    ucfg, bcfg = bench.full_configs()
    h = w = 64
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    on_dev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    usd, bsd = bench.synth_weights()
    for name in ("random weights", "zero matrices"):
        if name == "zero matrices":
            for sd in (usd, bsd):
                for k, v in sd.items():
                    if v.ndim > 1:
                        v.zero_()
        pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")

        def edit():
            return pipe(on_dev["prompt"], on_dev["fg"], on_dev["bg"], score, on_dev["dino"], num_inference_steps=50, latents=on_dev["latents"])
        edit()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            out = edit()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 4 / 50 * 1e3:.3f} ms per step (finite: {bool(torch.isfinite(out).all())})", flush=True)
        del pipe
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
