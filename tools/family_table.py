#!/usr/bin/env python3
"""Per-family table of a profiled configuration: convolution / row-chain / GEMM / attention - serial HIP-event
time of one BlobNet-active step (bench.py --table under tools/profile_round.sh), algorithmic GFLOP, TFLOP/s, and measured HBM + fabric
traffic (rocprofv3 PMC passes, 2 * FETCH + WRITE) against the algorithmic bytes.   python tools/family_table.py  ->  profiles/r6_family_table.md"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAGS = [("r6", "512^2 batch 1 (headline)"), ("r6_b2", "512^2 batch 2 (script default shape)"), ("r6_c3", "512^2 batch 8 (C3 / C4 per GPU)"), ("r6_c5", "768^2 batch 4 (C5)")]


def family(k):
    return ("convolution (conv_wreg / conv_halo)" if k.startswith("conv_") else "attention" if k.startswith("attn") else "row-chain" if k.startswith("rowchain") else
            "large-M GEMM (gemm256)" if k.startswith("gemm256") else "GEMM (gemm_fast / gemm_wreg + split-K reducers)" if k.startswith("gemm") or "splitk" in k else "other")


def main():
    out = ["# Round 6: per-family table of one BlobNet-active step (serial replay, HIP events around every launch; ~11 us of event overhead per launch included)", "",
           "`ms`, `GFLOP`, `TFLOP/s`: `profiles/<tag>_event_table.json` (bench.py --table).  `traffic / algorithmic`: HBM + fabric bytes of the family's kernels per step "
           "(`<tag>_pmc_hbm_traffic.json`, 2 x FETCH_SIZE + WRITE_SIZE, launches of the PMC pass scaled to one step) over the algorithmic bytes of its launches.", ""]
    for tag, title in TAGS:
        ev = json.load(open(os.path.join(REPO, "profiles", f"{tag}_event_table.json")))
        pmc = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_hbm_traffic.json")))
        steps = pmc.get("denoise_steps_profiled") or 4
        traffic = {}
        for r in pmc["kernels"]:
            name = r["kernel"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            for pre, short in (("_ZN12_GLOBAL__N_115attn_fwd_kernel", "attn_fwd_kernel"),):
                if name.startswith(pre):
                    name = short
            traffic[family(name.split("(")[0])] = traffic.get(family(name.split("(")[0]), 0.0) + r["hbm_bytes_per_launch_corrected"] * r["launches"] / steps
        fam = {}
        for k, v in ev["by_kernel"].items():
            f = fam.setdefault(family(k), dict(ms=0.0, gflop=0.0, bytes=0.0, n=0))
            f["ms"] += v["ms"]; f["gflop"] += v["tflops"] * v["ms"]; f["bytes"] += v["gbps"] * v["ms"] * 1e6; f["n"] += v["launches"]
        tot = sum(f["ms"] for f in fam.values())
        out += [f"## {title}  (`profiles/{tag}_*`, kernel sources {pmc.get('csrc_sha')}; serial sum {tot:.2f} ms)", "",
                "| family | launches | ms | share | GFLOP | TFLOP/s | traffic MB | algorithmic MB | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|---|"]
        for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
            tr = traffic.get(name, 0.0)
            if name == "other":            # (the PMC pass also sees the process's once-only torch kernels - weight upload, input copies: no per-step figure)
                out.append(f"| other (norm / layout / scheduler launches) | {f['n']} | {f['ms']:.2f} | {100 * f['ms'] / tot:.0f} % | - | - | - | {f['bytes'] / 1e6:.0f} | - |")
                continue
            out.append(f"| {name} | {f['n']} | {f['ms']:.2f} | {100 * f['ms'] / tot:.0f} % | {f['gflop']:.0f} | {f['gflop'] / max(f['ms'], 1e-9):.0f} | {tr / 1e6:.0f} | "
                       f"{f['bytes'] / 1e6:.0f} | {(tr / f['bytes']) if f['bytes'] else float('nan'):.2f} |")
        out.append("")
    open(os.path.join(REPO, "profiles", "r6_family_table.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
