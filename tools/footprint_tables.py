#!/usr/bin/env python3
"""Tile tables that trade a launch's latency against its CU footprint (round 3 experiment).

tools/tune_gemm.py picks, per GEMM shape, the candidate with the lowest stand-alone latency - which is usually the one that
spreads over the most CUs.  Inside the step two branches (BlobNet, UNet) share the device, so a candidate that takes 10 % longer
on half the CUs may be the better citizen.  From the tuner's --dump this writes tables minimising  us * (CUs / 256) ** beta.
Usage: python tools/footprint_tables.py dump.json outdir 0.25 0.5 ..."""
import json
import os
import sys

WPC = {1: 1, 2: 1, 3: 2, 4: 2, 5: 1, 6: 2, 7: 2}          # workgroups per CU by LDS footprint of the tile configuration


def main():
    dump = json.load(open(sys.argv[1]))
    outdir = sys.argv[2]
    os.makedirs(outdir, exist_ok=True)
    for beta in [float(b) for b in sys.argv[3:]]:
        table = {}
        changed = 0
        for shape, cands in dump.items():
            def cost(c):
                us, cfg, sk, nblk = c
                cus = min(256, -(-nblk // WPC[cfg]))
                return us * (cus / 256.0) ** beta
            best = min(cands, key=cost)
            fastest = min(cands, key=lambda c: c[0])
            changed += best[1:3] != fastest[1:3]
            table[shape] = [best[1], best[2]]
        path = os.path.join(outdir, f"tuning_beta{beta:g}.json")
        json.dump(dict(device="MI355X gfx950", beta=beta, shapes=table), open(path, "w"), indent=0, sort_keys=True)
        print(f"beta {beta:g}: {changed} of {len(table)} shapes differ from the latency-optimal choice -> {path}")


if __name__ == "__main__":
    main()
