"""Soak: the same full-size 50-step edit N times (graph replays; plus a second, per-request batch-2 plan in between to move the
allocator and the caches) - every repeat must reproduce the first result bit for bit.  A race in a kernel (counted waits, the halo
kernel's half-tap stagger, split-K slabs) would show up here as a nondeterministic output.  Usage (GPU): python tools/soak.py [N]"""
import os
import sys
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import splat_features
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    ucfg, bcfg = bench.full_configs()
    usd, bsd = bench.synth_weights()
    h = w = 64
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="unipc")

    def edit(steps=50):
        return pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=steps, latents=inp["latents"],
                    blobnet_control_guidance_end=0.9).cpu()
    first = edit()
    assert torch.isfinite(first).all()
    bad = 0
    for i in range(n):
        if i % 5 == 4:
            edit(7)                                   # another plan in between
        out = edit()
        if not torch.equal(out, first):
            bad += 1
            print(f"repeat {i}: DIFFERS, max abs {float((out - first).abs().max()):.3e}", flush=True)
    print(f"{n} repeats of a 50-step UniPC edit: {n - bad} bit-identical, {bad} different", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
