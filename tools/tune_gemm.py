#!/usr/bin/env python3
"""Measure every (tile configuration, split-K) candidate for the distinct GEMM shapes of the hot path on the MI355X the
script runs on, and write blobctrl_amd/gemm_tuning.json (consumed by launch.Recorder.plan_gemm).

Usage (GPU box):  python tools/tune_gemm.py [--res 512] [--out blobctrl_amd/gemm_tuning.json]
The shapes are collected by planning one BlobNet-active denoise step of the full-size model (no weights needed:
planning only records launches), so the table follows the engine automatically.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from blobctrl_amd import _lib  # noqa: E402
from blobctrl_amd.launch import Recorder  # noqa: E402


def collect_shapes(res):
    """Distinct (mode, M, N, K) with conv geometry, from the launch metadata of a planned step (random tiny weights are not
    needed: we synthesise the list from the known layer table of SURVEY Appendix A instead of allocating 3.4 GB)."""
    h = res // 8
    shapes = {}
    boc = (320, 640, 1280, 1280)

    def add(mode, M, N, K, conv=None):
        shapes.setdefault((mode, M, N, K), conv)

    for B in (1, 2):                                   # BlobNet runs at batch 1, the UNet at CFG batch 2
        H, W = h, 2 * h
        lv = [(H >> i, W >> i) for i in range(4)]
        for i, c in enumerate(boc):
            hh, ww = lv[i]
            M = B * hh * ww
            cins = {c, boc[max(i - 1, 0)], 2 * c, c + boc[max(i - 1, 0)], c + boc[min(i + 1, 3)]}
            for cin in cins:
                add("conv1", M, c, 9 * cin, dict(Cin=cin, Hin=hh, Win=ww, Hout=hh, Wout=ww, stride=1))
                add("dense", M, c, cin)
            add("dense", M, c, c)
            add("dense", M, 2 * c, c)
            add("dense", M, 8 * c, c)
            add("dense", M, c, 4 * c)
            if i < 3:
                add("conv2", B * (hh // 2) * (ww // 2), c, 9 * c, dict(Cin=c, Hin=hh, Win=ww, Hout=hh // 2, Wout=ww // 2, stride=2))
                hu, wu = lv[i + 1]
                cu = boc[i + 1]
                add("ups", M, cu, 9 * cu, dict(Cin=cu, Hin=hu, Win=wu, Hv=hh, Wv=ww, Hout=hh, Wout=ww, stride=1))
    return shapes


def time_launch_cold(rec, seg, stream, iters, thrash, touch):
    """In-situ-like timing: before every timed launch the caches are flushed by streaming `thrash` (> the 256 MiB Infinity Cache)
    and the activation operand is touched again (in the loop it was just written by its producer, so it sits in L2 / MALL while
    the weights - 3.4 GB per step - always come cold from HBM)."""
    lib = rec.lib
    tot = 0.0
    for it in range(iters + 1):
        thrash.add_(1)
        for t in touch:
            t.mul_(1)
        a, b = C.c_void_p(), C.c_void_p()
        lib.bc_event_create(C.byref(a)); lib.bc_event_create(C.byref(b))
        lib.bc_event_record(a, stream)
        seg.run(stream)
        lib.bc_event_record(b, stream)
        ms = C.c_float()
        _lib.check(lib.bc_event_elapsed_ms(a, b, C.byref(ms)), "elapsed")
        lib.bc_event_destroy(a); lib.bc_event_destroy(b)
        if it > 0:
            tot += ms.value
    return tot * 1e3 / iters


def time_launch(rec, seg, stream, iters):
    lib = rec.lib
    a, b = C.c_void_p(), C.c_void_p()
    lib.bc_event_create(C.byref(a)); lib.bc_event_create(C.byref(b))
    for _ in range(2):
        seg.run(stream)
    lib.bc_event_record(a, stream)
    for _ in range(iters):
        seg.run(stream)
    lib.bc_event_record(b, stream)
    ms = C.c_float()
    _lib.check(lib.bc_event_elapsed_ms(a, b, C.byref(ms)), "elapsed")
    lib.bc_event_destroy(a); lib.bc_event_destroy(b)
    return ms.value * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--out", default=os.path.join(REPO, "blobctrl_amd", "gemm_tuning.json"))
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--gn-penalty-us", type=float, default=0.0,
                    help="extra cost charged to split-K on convolutions (0: the split-K reducer emits the GroupNorm partials too)")
    ap.add_argument("--dump", default=None, help="also write EVERY candidate's time (us, cfg, splitk, workgroups) per shape to this JSON")
    ap.add_argument("--only", default=None, help="regular expression on 'mode|M|N|K': measure these shapes only and MERGE them into --out")
    ap.add_argument("--cold", action="store_true", help="flush the caches before every timed launch (weights cold, activations warm)")
    args = ap.parse_args()
    os.environ["BC_NO_TUNING"] = "1"
    dev = torch.device("cuda:0")
    rec = Recorder(dev)
    stream = torch.cuda.current_stream().cuda_stream
    shapes = collect_shapes(args.res)
    if args.res == 512:
        shapes.update(collect_shapes(768)) if os.environ.get("BC_TUNE_768") else None
    table, report, dump = {}, [], {}
    t0 = time.time()
    thrash = torch.zeros(160 * 1024 * 1024, dtype=torch.float32, device=dev) if args.cold else None      # 640 MB
    if args.only:
        import re
        shapes = {k: v for k, v in shapes.items() if re.search(args.only, "|".join(str(x) for x in k))}
        if os.path.exists(args.out):
            with open(args.out) as f:
                table.update(json.load(f).get("shapes", {}))
    for (mode, M, N, K), conv in sorted(shapes.items()):
        nk = K // 64
        if K % 64 or (conv and conv["Cin"] % 64):
            continue
        Cin = conv["Cin"] if conv else K
        A = torch.randn(M if not conv else (M // (conv["Hout"] * conv["Wout"])) * conv["Hin"] * conv["Win"], Cin,
                        device=dev, dtype=torch.float16)
        Wt = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        bias = torch.zeros(N, device=dev)
        best = None
        results = []
        for cfg in range(1, 8):
            for sk in (1, 2, 3, 4, 6, 8, 12):
                if sk > 1 and nk // sk < 4:
                    continue
                bm = [0, 256, 128, 128, 256, 256, 128, 64][cfg]
                bn = [0, 128, 128, 128, 64, 64, 64, 64][cfg]
                nblk = -(-M // bm) * -(-N // bn) * sk
                if sk > 1 and nblk > 1024:
                    continue
                if nblk > 16384:
                    continue
                seg = rec.begin("t")
                rec.gemm(A=A, W=Wt, M=M, N=N, K=K, out=out, bias=bias, conv=conv, tile_cfg=cfg, splitk=sk,
                         rows_per_batch=(conv["Hout"] * conv["Wout"] if conv else 0))
                us = time_launch_cold(rec, seg, stream, args.iters, thrash, (A,)) if args.cold else \
                    time_launch(rec, seg, stream, args.iters)
                results.append((us, cfg, sk))
                dump.setdefault(f"{mode}|{M}|{N}|{K}", []).append([round(us, 2), cfg, sk, nblk])
                cost = us + (args.gn_penalty_us if (sk > 1 and mode != "dense") else 0.0)
                if best is None or cost < best[3]:
                    best = (us, cfg, sk, cost)
        auto_cfg, auto_sk, _, _ = rec.plan_gemm(M, N, K, True, mode)
        auto_us = [r[0] for r in results if r[1] == auto_cfg and r[2] == auto_sk]
        table[f"{mode}|{M}|{N}|{K}"] = [best[1], best[2]]
        tf = 2.0 * M * N * K / best[0] / 1e6
        report.append(dict(shape=[mode, M, N, K], best=_lib.TILE_NAMES[best[1]], splitk=best[2], us=round(best[0], 1),
                           tflops=round(tf, 1), heuristic=_lib.TILE_NAMES[auto_cfg], heuristic_sk=auto_sk,
                           heuristic_us=round(auto_us[0], 1) if auto_us else None))
        print(report[-1], flush=True)
    with open(args.out, "w") as f:
        json.dump(dict(device="MI355X gfx950", res=args.res, shapes=table), f, indent=0, sort_keys=True)
    if args.dump:
        with open(args.dump, "w") as f:
            json.dump(dump, f)
    print(f"wrote {len(table)} shapes to {args.out} in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
