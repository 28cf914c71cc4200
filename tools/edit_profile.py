"""cProfile of the engine's per-edit front end (inputs on the host, plan and graphs already built): where the host time of an edit
goes besides waiting for the GPU."""
import cProfile
import os
import pstats
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
    usd, bsd = bench.synth_weights()
    h = w = 64
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")
    on_dev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}

    def edit(src):
        return pipe(src["prompt"], src["fg"], src["bg"], score, src["dino"], num_inference_steps=50, latents=src["latents"])
    for src, name in ((inp, "host inputs"), (on_dev, "device inputs")):
        edit(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            edit(src)
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per edit", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        edit(on_dev)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
