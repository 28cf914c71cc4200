"""Where a denoise step's time goes: graph-replayed active step (BlobNet + UNet, two streams), inactive step (UNet only),
and both on ONE stream (serial).  Uses bench.py's synthetic full-size setup."""
import os
import sys
import time
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                        # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import splat_features
    dev = torch.device("cuda:0")
    ucfg, bcfg = bench.full_configs()
    usd, bsd = bench.synth_weights()
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    h = w = res // 8
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    order = [x == "1" for x in (sys.argv[2] if len(sys.argv) > 2 else "01")]
    for one_stream in order:
        if one_stream:
            os.environ["BC_ONE_STREAM"] = "1"
        else:
            os.environ.pop("BC_ONE_STREAM", None)
        pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")
        pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=4, latents=inp["latents"])
        P = pipe.plan_for(1, h, w, 77, 768, 4)
        s, side = pipe._streams()

        def run(seg):
            with torch.cuda.stream(pipe.stream):          # same stream as the replay: the counter must not run past the tables
                P.step_idx.zero_()
            seg.run(s, side, pipe._extra())
        print(f"one_stream={one_stream}: active step {timeit(lambda: run(P.step_active)):.3f} ms, "
              f"inactive (UNet only) {timeit(lambda: run(P.step_inactive)):.3f} ms; launches active "
              f"{len(P.step_active)} inactive {len(P.step_inactive)}", flush=True)
        del pipe, P
        if not os.environ.get("PROBE_NO_GC"):
            import gc
            gc.collect()
        if not os.environ.get("PROBE_NO_EMPTY"):
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
