#!/usr/bin/env python3
"""Compile the tiny-net edit of tests/golden/loop_tiny.npz (DDIM, 5 steps, guidance window [0, 1]) into a `.bcplan` file and write
its inputs / expected result as a flat binary for the plain-C host test (tests/c/plan_edit.c).  Runs WITHOUT a GPU: the plan is
compiled against host addresses and is relocatable.  Weights come from blobctrl_amd.synth (same seeds as every tiny fixture).

    <out>/tiny_edit.bcplan            plan: segments prologue / step_active / step_inactive, weights + scheduler tables embedded
    <out>/tiny_edit_io.bin            u32 count, then per record: char name[32], u64 bytes, data  (inputs named like the plan's
                                      buffers; "expected_latents" = the reference loop's final latents; "sequence" = int32 per step,
                                      1 = step_active)
"""
import os
import struct
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from blobctrl_amd.pipeline import BlobCtrlEngine  # noqa: E402
from tests.common import TINY, g, tiny_weights  # noqa: E402
from tests.gpu_common import tiny_trunk_configs  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "plan_fixture")


def main():
    os.makedirs(OUT, exist_ok=True)
    z = np.load(os.path.join(GOLD, "loop_tiny.npz"))
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_trunk_configs()
    eng = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device="cpu", scheduler="ddim", compile_only=True)
    B, h, w, T, steps = 1, 8, 8, 7, 5
    seq = eng.compile_plan(os.path.join(OUT, "tiny_edit.bcplan"), B, h, w, T, TINY["ctx"], steps, guidance_scale=7.5,
                           blobnet_conditioning_scale=1.0, blobnet_control_guidance_start=0.0, blobnet_control_guidance_end=1.0)
    score = torch.from_numpy(z["gs_score"]).float()                     # [1,2,h,w] = (bg, fg)
    dino = g(35, 1, 1, TINY["feat"])
    feat16 = torch.zeros(1, 8, dtype=torch.float16)
    feat16[:, : TINY["feat"]] = dino.reshape(1, -1).half()
    recs = {
        "latents": g(31, B, 4, h, w).numpy(),                              # x init_noise_sigma (= 1 for DDIM)
        "ctx": g(32, 2 * B, T, TINY["ctx"]).half().numpy(),
        "fg_lat": (g(33, 1, 4, h, w) * 0.18215 * 5).numpy(),
        "bg_lat": (g(34, 1, 4, h, w) * 0.18215 * 5).numpy(),
        "bg_score": score[:, 0].contiguous().numpy(),
        "fg_score": score[:, 1].contiguous().numpy(),
        "feat": dino.reshape(1, -1).numpy(),
        "feat16": feat16.numpy(),
        "expected_latents": z["ddim_5_final"].astype(np.float32),
        "sequence": np.array([1 if s == "step_active" else 0 for s in seq], np.int32),
    }
    with open(os.path.join(OUT, "tiny_edit_io.bin"), "wb") as f:
        f.write(struct.pack("<I", len(recs)))
        for name, arr in recs.items():
            raw = np.ascontiguousarray(arr).tobytes()
            f.write(name.encode().ljust(32, b"\0"))
            f.write(struct.pack("<Q", len(raw)))
            f.write(raw)
    for fn in ("tiny_edit.bcplan", "tiny_edit_io.bin"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


if __name__ == "__main__":
    main()
