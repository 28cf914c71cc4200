#!/usr/bin/env python3
"""Timeline of ONE denoise step from a rocprofv3 --kernel-trace CSV (run on the GPU box by tools/timeline.sh): per-queue busy time,
idle gaps between consecutive kernels of a queue, overlap of the two queues, and the largest gaps with the kernels around them.
Usage: python3 tools/timeline.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    # steps are delimited by the scheduler kernel (cfg_step): take the last complete window between two of them
    marks = [i for i, r in enumerate(rows) if "cfg_step" in r[2]]
    if len(marks) < 3:
        print("not enough steps in trace", len(marks))
        return
    a, b = marks[-3], marks[-2]
    win = rows[a + 1:b + 1]
    t0, t1 = win[0][0], max(r[1] for r in win)
    span = (t1 - t0) / 1e3
    by_q = {}
    for r in win:
        by_q.setdefault(r[3], []).append(r)
    out = {"trace": os.path.basename(f), "kernels_in_step": len(win), "step_span_us": round(span, 1), "queues": {}}
    iv = []
    for q, rs in by_q.items():
        busy = sum(r[1] - r[0] for r in rs) / 1e3
        gaps = [(rs[i + 1][0] - rs[i][1], rs[i][2][:60], rs[i + 1][2][:60]) for i in range(len(rs) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        out["queues"][q] = {
            "kernels": len(rs), "busy_us": round(busy, 1), "first_start_us": round((rs[0][0] - t0) / 1e3, 1),
            "last_end_us": round((rs[-1][1] - t0) / 1e3, 1), "gap_sum_us": round(sum(g[0] for g in pos) / 1e3, 1),
            "gap_median_us": round(sorted(g[0] for g in pos)[len(pos) // 2] / 1e3, 2) if pos else 0,
            "gaps_over_10us": sum(1 for g in pos if g[0] > 10000),
            "largest_gaps": [dict(us=round(g[0] / 1e3, 1), after=g[1], before=g[2]) for g in sorted(pos, reverse=True)[:8]]}
        iv += [(r[0], r[1]) for r in rs]
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    depth, last, t_any, t_two = 0, ev[0][0], 0, 0
    for t, dlt in ev:
        if depth >= 1:
            t_any += t - last
        if depth >= 2:
            t_two += t - last
        depth += dlt
        last = t
    out["any_kernel_running_us"] = round(t_any / 1e3, 1)
    out["two_or_more_running_us"] = round(t_two / 1e3, 1)
    out["idle_us"] = round(span - t_any / 1e3, 1)
    agg = {}
    for r in win:
        k = r[2].split("(")[0][-70:]
        a_ = agg.setdefault(k, [0, 0.0])
        a_[0] += 1
        a_[1] += (r[1] - r[0]) / 1e3
    out["by_kernel_us"] = {k: dict(n=v[0], us=round(v[1], 1)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)


if __name__ == "__main__":
    main()
