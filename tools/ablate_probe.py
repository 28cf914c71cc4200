"""Marginal cost of each kernel family inside the two-stream, graph-replayed step: re-capture the active step WITHOUT the launches
of one kind (results are garbage, the timing is what matters) and compare with the full step."""
import os
import sys
import time
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                        # noqa: E402


def timeit(fn, n=15):
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
    h = w = 64
    inp = bench.synth_inputs(h, w, batch=1)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")
    pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=4, latents=inp["latents"])
    P = pipe.plan_for(1, h, w, 77, 768, 4)
    s, side = pipe._streams()
    full = P.step_active
    kinds = sorted({m["kind"] for m in full.meta})
    print("kinds:", {k: sum(1 for m in full.meta if m["kind"] == k) for k in kinds})

    def run(seg):
        with torch.cuda.stream(pipe.stream):
            P.step_idx.zero_()
        seg.run(s, side)
    base = timeit(lambda: run(full))
    print(f"full step: {base:.3f} ms")
    groups = {"attention": ("attention",), "groupnorm": ("groupnorm", "groupnorm_fused_stats", "gn_finalize"), "layernorm": ("layernorm",),
              "gn_finalize only": ("gn_finalize",), "transformer (all but attention)": ("layernorm", "ff", "qkv", "attn_out"),
              "self-attn L0 only": ("attention@8192",), "everything but conv3x3": ("attention", "groupnorm", "groupnorm_fused_stats",
              "gn_finalize", "layernorm", "ff", "qkv", "attn_out", "conv1x1", "zero_conv"),
              "ff": ("ff",), "conv3x3+conv_in+up/down": ("conv3x3", "conv_in", "upsample", "downsample", "conv_out"),
              "qkv+attn_out": ("qkv", "attn_out"), "conv1x1": ("conv1x1",), "zero_conv": ("zero_conv",), "temb": ("temb",)}
    for name, ks in groups.items():
        off = [i for i, m in enumerate(full.meta)
               if m["kind"] in ks or ("attention@8192" in ks and m["kind"] == "attention" and m["shape"][3] == 8192 and m["shape"][4] == 8192)]
        for i in off:
            full.enable(i, False)
        full.release()
        run(full)
        torch.cuda.synchronize()
        full.capture(s, side)
        t = timeit(lambda: run(full))
        print(f"without {name:26s} ({len(off):3d} launches): {t:7.3f} ms  -> marginal cost {base - t:6.3f} ms", flush=True)
        for i in off:
            full.enable(i, True)
    full.release()
    full.capture(s, side)


if __name__ == "__main__":
    main()
