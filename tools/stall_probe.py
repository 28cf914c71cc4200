"""How long does the UNet stream sit waiting for BlobNet residuals?  Eager two-stream replay of one active step with a HIP event
before and after every cross-stream wait on the main stream; prints the per-wait stall and the total."""
import ctypes as C
import os
import sys
import time
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                        # noqa: E402
from blobctrl_amd import _lib                                       # noqa: E402


def main():
    from blobctrl_amd.pipeline import StableDiffusionBlobNetPipeline
    from blobctrl_amd.splat import splat_features
    dev = torch.device("cuda:0")
    ucfg, bcfg = bench.full_configs()
    usd, bsd = bench.synth_weights()
    h = w = 64
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    pipe = StableDiffusionBlobNetPipeline(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")
    pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=4, latents=inp["latents"])
    P = pipe.plan_for(1, h, w, 77, 768, 4)
    s, side = pipe._streams()
    lib = _lib.load()
    seg = P.step_active
    streams = (s, side)

    def ev():
        e = C.c_void_p()
        _lib.check(lib.bc_event_create(C.byref(e)), "ev")
        return e
    for rep in range(3):
        with torch.cuda.stream(pipe.stream):
            P.step_idx.zero_()
        torch.cuda.synchronize()
        marks = []
        t0 = ev(); lib.bc_event_record(t0, s)
        last_main = None
        for i, (fn, sid, m) in enumerate(zip(seg.calls, seg.sids, seg.meta)):
            if m["kind"] == "event_wait" and sid == 0:
                a = ev(); lib.bc_event_record(a, s)
                fn(streams[sid])
                b = ev(); lib.bc_event_record(b, s)
                marks.append((i, a, b))
            else:
                fn(streams[sid])
        t1 = ev(); lib.bc_event_record(t1, s)
        torch.cuda.synchronize()
        ms = C.c_float()
        lib.bc_event_elapsed_ms(t0, t1, C.byref(ms))
        total = ms.value
        stalls = []
        for (i, a, b) in marks:
            lib.bc_event_elapsed_ms(a, b, C.byref(ms))
            stalls.append((i, ms.value))
        print(f"rep {rep}: eager two-stream step {total:.3f} ms (main stream), {len(stalls)} waits, total stall {sum(v for _, v in stalls):.3f} ms")
        if rep == 2:
            print("  per wait (launch index: ms):", ", ".join(f"{i}:{v:.3f}" for i, v in stalls))


if __name__ == "__main__":
    main()
