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
    # (round 5) predicates over the launch metadata: by op kind, by queue, by level (rows of the output), by kernel
    rows = lambda m: (m.get("shape") or (None, 0))[1] if m["kind"] != "attention" else 0
    variant = lambda m: m.get("variant") or ""
    groups = {
        "BlobNet queue (everything on stream 1)": lambda m: m["sid"] == 1,
        "attention": lambda m: m["kind"] == "attention",
        "attention D=40 (8192 tokens)": lambda m: m["kind"] == "attention" and m["shape"][2] == 40,
        "attention D=80 / 160": lambda m: m["kind"] == "attention" and m["shape"][2] in (80, 160),
        "row-chain (all)": lambda m: m["kind"] in ("rowchain", "rowchain_sum"),
        "row-chain 320 channels": lambda m: m["kind"] == "rowchain" and m["shape"][2] == 320,
        "row-chain 640 channels (+ sum)": lambda m: m["kind"] in ("rowchain", "rowchain_sum") and m["shape"][2] == 640,
        "conv3x3 (ResBlock convolutions)": lambda m: m["kind"] == "conv3x3",
        "conv3x3 at 64 x 128": lambda m: m["kind"] == "conv3x3" and rows(m) in (16384, 8192),
        "conv3x3 at 32 x 64": lambda m: m["kind"] == "conv3x3" and rows(m) in (4096, 2048),
        "conv3x3 at 16 x 32 and 8 x 16": lambda m: m["kind"] == "conv3x3" and rows(m) in (1024, 512, 256, 128),
        "up / down-sample convolutions, conv_in / out": lambda m: m["kind"] in ("upsample", "downsample", "conv_in", "conv_out"),
        "gemm_wreg projections (1280-channel levels)": lambda m: "gemm_wreg" in variant(m),
        "feed-forward of the 1280-channel levels": lambda m: m["kind"] == "ff",
        "attn2.to_q of the 1280-channel blocks (UNet)": lambda m: m["kind"] == "qkv" and (m.get("shape") or ("",))[0] == "gw_ln",
        "cross-attention launches (77 keys, D = 160)": lambda m: m["kind"] == "attention" and m["shape"][4] == 77,
        "split-K convolutions only (16 x 32, 8 x 16) - reducers included": lambda m: m["kind"] == "conv3x3" and "splitk" in variant(m),
        "1x1 convolutions (shortcuts, proj_in / out)": lambda m: m["kind"] == "conv1x1",
        "zero-convs": lambda m: m["kind"] == "zero_conv",
        "GroupNorm passes + statistics": lambda m: m["kind"] in ("groupnorm", "groupnorm_fused_stats", "gn_finalize", "gn_stats", "memset"),
    }
    out = {}
    for name, pred in groups.items():
        off = [i for i, m in enumerate(full.meta) if m["kind"] not in ("event_record", "event_wait", "assemble", "cfg_step") and pred(m)]
        for i in off:
            full.enable(i, False)
        full.release()
        run(full)
        torch.cuda.synchronize()
        full.capture(s, side)
        t = timeit(lambda: run(full))
        out[name] = dict(launches=len(off), step_ms_without=round(t, 3), marginal_ms=round(base - t, 3))
        print(f"without {name:46s} ({len(off):3d} launches): {t:7.3f} ms  -> marginal cost {base - t:6.3f} ms", flush=True)
        for i in off:
            full.enable(i, True)
    if os.environ.get("ABLATE_DUMP"):
        import json
        with open(os.environ["ABLATE_DUMP"], "w") as f:
            json.dump(dict(full_step_ms=round(base, 3), without=out), f, indent=1)
    full.release()
    full.capture(s, side)


if __name__ == "__main__":
    main()
