#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC counter CSVs per kernel (run on the GPU box by tools/profile_round.sh) and, in the build container,
copy the round's summaries into profiles/."""
import collections
import csv
import glob
import json
import os
import sys


def agg(path_glob):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for p in glob.glob(path_glob, recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                k = out[r["Kernel_Name"]][r["Counter_Name"]]
                k[0] += float(r["Counter_Value"])
                k[1] += 1
    return out


def csrc_sha():
    """Same hash as bench.py: identifies the kernel sources these counters were measured on."""
    import hashlib
    hsh = hashlib.sha1()
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blobctrl_amd", "csrc")
    for f in sorted(x for x in os.listdir(src) if x.endswith((".hip", ".h"))):
        with open(os.path.join(src, f), "rb") as fh:
            hsh.update(f.encode() + b"\0" + fh.read())
    return hsh.hexdigest()[:12]


def workload_of(d):
    """(res, batch) of the bench run the passes profiled, from the metric name of its line (bench.metric_name)."""
    import re
    try:
        with open(os.path.join(d, "stats_bench.json")) as f:
            metric = json.loads(f.read().strip().splitlines()[-1])["metric"]
        m = re.match(r"(\d+)x\d+_", metric)
        b = re.search(r"_batch(\d+)", metric)
        return int(m.group(1)), int(b.group(1)) if b else 1
    except Exception:                                               # noqa: BLE001
        return 512, 1


def main():
    d = sys.argv[1]
    res, batch = workload_of(d)
    fetch, write, mfma = (agg(os.path.join(d, sub, "**", "*counter_collection.csv")) for sub in ("pmc_fetch", "pmc_write", "pmc_mfma"))
    kernels = []
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, {}).get("FETCH_SIZE", [0.0, 0])
        w, nw = write.get(name, {}).get("WRITE_SIZE", [0.0, 0])
        n = max(nf, nw, 1)
        kernels.append(dict(kernel=name, launches=n, fetch_kib_raw=round(f / max(nf, 1), 1), write_kib_raw=round(w / max(nw, 1), 1),
                            hbm_bytes_per_launch_corrected=int((2 * f / max(nf, 1) + w / max(nw, 1)) * 1024)))
    kernels.sort(key=lambda r: -r["hbm_bytes_per_launch_corrected"] * r["launches"])
    json.dump(dict(command="rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 1 --warmup 1 "
                           "--no-cpu-baseline --no-roofline --denoise-steps 2",
                   correction="bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports half of a wide coalesced read "
                              "stream; WRITE_SIZE exact; MI355X_MICROARCH.md section HBM)",
                   # denoise steps the pass executed = launches of the per-step CFG / scheduler kernel (two 2-step edits + the warm-up
                   # replays of the active and the inactive segment before capture): per-step traffic = sum over the library's kernels / this
                   denoise_steps_profiled=max([k["launches"] for k in kernels if "cfg_step_kernel" in k["kernel"]] or [4]),
                   csrc_sha=csrc_sha(), res=res, batch=batch, kernels=kernels), open(os.path.join(d, "pmc_hbm_traffic.json"), "w"), indent=1)
    util = []
    for name, c in mfma.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        n = c["GRBM_GUI_ACTIVE"][1]
        cyc = c["GRBM_GUI_ACTIVE"][0] / 8.0                      # summed over the 8 XCDs
        util.append(dict(kernel=name, launches=n, gpu_cycles_per_launch=round(cyc / n, 1),
                         mfma_busy_frac=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 1])[0] / (1024.0 * cyc), 4),
                         valu_issue_busy_frac=round(c.get("SQ_ACTIVE_INST_VALU", [0, 1])[0] * 4 / (1024.0 * cyc), 4),
                         lds_busy_frac=round(c.get("SQ_LDS_IDX_ACTIVE", [0, 1])[0] / (256.0 * cyc), 4)))
    util.sort(key=lambda r: -r["gpu_cycles_per_launch"] * r["launches"])
    json.dump(dict(command="rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -- python3 bench.py "
                           "--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --denoise-steps 2",
                   note="fractions of the launch's own GPU cycles (GRBM_GUI_ACTIVE/8): MFMA pipe busy over 1024 SIMDs, VALU issue port "
                        "(SQ_ACTIVE_INST_VALU counts 4-cycle quads), LDS array over 256 CUs; all launches of a kernel summed",
                   csrc_sha=csrc_sha(), res=res, batch=batch, kernels=util), open(os.path.join(d, "pmc_mfma_util.json"), "w"), indent=1)
    print("summaries:", os.listdir(d))


def publish(src, tag):
    """Build container: copy the summaries of gpurun_out/prof into profiles/<tag>_* and write <tag>_kernel_summary.md."""
    import shutil
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(repo, "profiles")
    shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_bench_steps1.csv"))
    shutil.copy(os.path.join(src, "pmc_hbm_traffic.json"), os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"))
    shutil.copy(os.path.join(src, "pmc_mfma_util.json"), os.path.join(dst, f"{tag}_pmc_mfma_util_per_kernel.json"))
    shutil.copy(os.path.join(src, "stats_bench.json"), os.path.join(dst, f"{tag}_bench_under_rocprofv3.json"))
    traffic = {r["kernel"]: r for r in json.load(open(os.path.join(src, "pmc_hbm_traffic.json")))["kernels"]}
    util = {r["kernel"]: r for r in json.load(open(os.path.join(src, "pmc_mfma_util.json")))["kernels"]}
    sha = json.load(open(os.path.join(src, "pmc_hbm_traffic.json")))["csrc_sha"]
    rows = list(csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))))
    lines = [f"# {tag} per-kernel summary (one MI355X, `bench.py --steps 1 --warmup 1`, 2 edits x 50 DDIM steps under rocprofv3; kernel sources {sha})", "",
             f"Sources: `{tag}_rocprofv3_kernel_stats_bench_steps1.csv` (durations), `{tag}_pmc_hbm_traffic.json` (HBM/fabric bytes per launch = "
             "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate PMC passes on a 2-step edit), "
             f"`{tag}_pmc_mfma_util_per_kernel.json` (SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE/8)).  Produced by "
             "`tools/profile_round.sh` + `tools/summarize_profile.py --publish`.", "",
             "| kernel | calls | avg us | % of kernel time | HBM+fabric MB / launch | GB/s | MFMA busy | VALU issue | LDS |", "|---|---|---|---|---|---|---|---|---|"]
    for r in rows[:28]:
        name = r["Name"]
        t, u = traffic.get(name), util.get(name)
        avg = float(r["AverageNs"]) / 1e3
        mb = t["hbm_bytes_per_launch_corrected"] / 1e6 if t else float("nan")
        short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
        short = short.split("(")[0][:64]
        lines.append(f"| `{short}` | {r['Calls']} | {avg:.1f} | {float(r['Percentage']):.2f} | {mb:.1f} | {mb * 1e3 / avg:.0f} | "
                     + (f"{100 * u['mfma_busy_frac']:.0f} % | {100 * u['valu_issue_busy_frac']:.0f} % | {100 * u['lds_busy_frac']:.0f} % |" if u else "- | - | - |"))
    open(os.path.join(dst, f"{tag}_kernel_summary.md"), "w").write("\n".join(lines) + "\n")
    print("published", sorted(f for f in os.listdir(dst) if f.startswith(tag)))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--publish":
        publish(sys.argv[2], sys.argv[3])
    else:
        main()
