#!/usr/bin/env python3
"""A/B probe of the fused row-chain launches (csrc/rowchain.hip) against the unfused launch list they replace, on the production
shapes of the 64 x 128 level (UNet: B = 2, 16384 rows; BlobNet: B = 1, 8192 rows), one process, interleaved, attention excluded
(both arms call the same attention kernel).  Reports microseconds per transformer block and the TFLOP/s of the fused part.
PROBE_COLD=1 flushes the caches before every timed replay (weights from HBM, as inside the step)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from tools.tune_gemm import time_launch, time_launch_cold  # noqa: E402


def build(fused, cross, B, H, W, zero, C=320):
    from blobctrl_amd import synth
    from blobctrl_amd.engine import Act, TrunkConfig, TrunkPlan
    from blobctrl_amd.launch import Recorder, encode_gn_tot
    from blobctrl_amd.weights import PackedTrunk
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from tests.common import block_param_shapes
    if fused:
        os.environ.pop("BC_PLAN", None)
    else:
        os.environ["BC_PLAN"] = "rowchain=0"
    sd = {"blk." + k: v for k, v in synth.synth_state_dict(block_param_shapes("transformer", dict(C=C, ctx=768 if cross else None)), 7).items()}
    sd["conv_in.weight"] = torch.zeros(8, 4, 3, 3)
    sd["none.time_emb_proj.weight"], sd["none.time_emb_proj.bias"] = torch.zeros(8, 1280), torch.zeros(8)
    if zero:
        sd["blk.zero.weight"], sd["blk.zero.bias"] = torch.randn(C, C, 1, 1) * 0.05, torch.zeros(C)
    dev = torch.device("cuda:0")
    pw = PackedTrunk(sd, dev, (320, 640, 1280, 1280))
    cfg = TrunkConfig(in_channels=4, num_heads=8, norm_num_groups=32, cross_attention_dim=768 if cross else None)
    rec = Recorder(dev)
    seg = rec.begin("blk")
    plan = TrunkPlan(rec, pw, cfg, B, H, W)
    plan.res_events, plan.res_bmod = None, 1
    x = torch.randn(B, H * W, C, device=dev, dtype=torch.float16)
    ns = H * W // 128
    f = x.float().view(B, ns, 128, C)
    rec.tots[x.data_ptr()] = encode_gn_tot(torch.stack([f.sum((1, 2)), (f * f).sum((1, 2))], -1)).to(x.device)      # producer statistics, as in the step
    if cross:
        plan.record_context(torch.randn(B * 77, 768, device=dev, dtype=torch.float16), 77)
    r2 = (torch.randn(1, H * W, C, device=dev) * 0.5).half() if cross else None
    zspec = ("blk.zero", 1.0, torch.ones(4, device=dev), torch.zeros(1, dtype=torch.int32, device=dev), 0) if zero else None
    n0 = len(seg.meta)
    out, pre = plan.transformer("blk.", Act(x, C, H, W), r2=r2, zero=zspec)
    if zero and pre is None:
        plan.dense(out.t, B * H * W, C, "blk.zero", C, kind="zero_conv", alpha=1.0, alpha_dev=zspec[2], alpha_idx=zspec[3], alpha_bstride=0,
                   rows_per_batch=H * W)
    for i, m in enumerate(seg.meta):                        # attention is common to both arms: leave it out of the timing
        if m["kind"] in ("attention", "ctx_kv"):
            seg.enable(i, False)
    flops = sum(m["flops"] for m in seg.meta if m["kind"] not in ("attention", "ctx_kv"))
    launches = sum(1 for m in seg.meta if m["kind"] not in ("attention", "ctx_kv", "event_wait", "event_record"))
    return rec, seg, x, flops, launches


def main():
    dev = torch.device("cuda:0")
    cold = bool(os.environ.get("PROBE_COLD"))
    thrash = torch.zeros(160 << 20, dtype=torch.float32, device=dev) if cold else None
    stream = torch.cuda.current_stream().cuda_stream
    Cc = int(os.environ.get("PROBE_C", "320"))                 # 640: the 32 x 64 level (BC_PLAN rowchain_min_blocks_640=1 to fuse BlobNet's too)
    Hh, Ww = (64, 128) if Cc == 320 else (32, 64)
    for name, cross, B, zero in ((f"UNet  B=2 ({2 * Hh * Ww} rows)", True, 2, False), (f"BlobNet B=1 ({Hh * Ww} rows)", False, 1, True)):
        arms = {}
        for fused in (False, True):
            arms["row-chain" if fused else "unfused"] = build(fused, cross, B, Hh, Ww, zero, Cc)
        res = {}
        for rnd in range(3):
            for k, (rec, seg, x, flops, n) in arms.items():
                us = time_launch_cold(rec, seg, stream, 6, thrash, [x]) if cold else time_launch(rec, seg, stream, 10)
                res.setdefault(k, []).append(us)
        line = f"{name} {'cold' if cold else 'hot'}:"
        for k, (rec, seg, x, flops, n) in arms.items():
            us = sorted(res[k])[1]
            line += f" | {k}: {n} launches {us:7.1f} us ({flops / us / 1e6:5.0f} TF/s)"
        print(line, flush=True)
        rec, seg, x, flops, n = arms["row-chain"]
        rows = seg.run_timed_kernels(stream)
        print("      " + " | ".join(f"{(m['variant'] or m['kind'])} {a * 1e3:.1f} us" for m, a, r in rows if a > 0.002), flush=True)


if __name__ == "__main__":
    main()
