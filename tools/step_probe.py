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
        def run_eager(seg):
            with torch.cuda.stream(pipe.stream):
                P.step_idx.zero_()
            seg.release()                                   # back to the launch list: bc_step enqueues every kernel itself
            seg.run(s, side, pipe._extra())
        if not one_stream and os.environ.get("PROBE_SUSTAINED"):
            for n in (20, 50, 150, 300):
                print(f"active step, {n} back-to-back graph replays: {timeit(lambda: run(P.step_active), n):.3f} ms each", flush=True)
            lat = inp["latents"]
            def edit():
                pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=50, latents=lat)
            edit()
            print(f"whole 50-step edit through the engine (one graph): {timeit(edit, 3) / 50:.3f} ms per step", flush=True)
            P50 = pipe.plan_for(1, h, w, 77, 768, 50)
            s50, side50 = pipe._streams()

            def manual():
                with torch.cuda.stream(pipe.stream):
                    P50.step_idx.zero_()
                    P50.hist.zero_()
                P50.prologue.run(s50)
                for _ in range(50):
                    P50.step_active.run(s50, side50, pipe._extra())
            print(f"prologue + 50 per-step graph replays, no engine front end: {timeit(manual, 3) / 50:.3f} ms per step", flush=True)
            g = next(iter(P50.loop_graphs.values()))

            def loop_only():
                with torch.cuda.stream(pipe.stream):
                    P50.step_idx.zero_()
                    P50.hist.zero_()
                pipe.lib.bc_graph_launch(g, s50)
            print(f"the whole-edit graph alone: {timeit(loop_only, 3) / 50:.3f} ms per step", flush=True)

            def prologue_only():
                P50.prologue.run(s50)
            print(f"prologue segment alone: {timeit(prologue_only, 10):.3f} ms", flush=True)
            # host time of the launch calls themselves (no synchronisation inside the timed region)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(P.step_active)
            t_host = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t0) * 1e3
            print(f"one step graph: hipGraphLaunch returns after {t_host:.3f} ms on the host, GPU done after {t_all:.3f} ms", flush=True)
            t0 = time.perf_counter()
            loop_only()
            t_host = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t0) * 1e3
            print(f"whole-edit graph: hipGraphLaunch returns after {t_host:.1f} ms on the host, GPU done after {t_all:.1f} ms", flush=True)
            P.step_active.release()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(P.step_active)
            t_host = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t0) * 1e3
            print(f"one step EAGER (727 launches from the C loop): host {t_host:.3f} ms, GPU done after {t_all:.3f} ms", flush=True)
            P.step_active.capture(s, side, pipe._extra())
        if not one_stream and os.environ.get("PROBE_EAGER"):
            t_graph = timeit(lambda: run(P.step_active))
            t_eager = timeit(lambda: run_eager(P.step_active))
            print(f"active step: graph replay {t_graph:.3f} ms, eager two-stream launches {t_eager:.3f} ms", flush=True)
            P.step_active.capture(s, side, pipe._extra())
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
