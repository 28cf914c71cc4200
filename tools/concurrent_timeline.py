"""Where the two queues REALLY are inside an overlapped step (GPU).

tools/critical_path.py replays the two queues on paper from isolated per-launch durations; this tool measures the overlap itself: the
active step is replayed eagerly on its two streams (`bc_plan_run_marked`) with a timestamp behind every STRIDE-th launch of each queue
(a few dozen events per replay: no measurable perturbation, where an event behind every launch would add ~5 us each), and every
interval between two marks of a queue is set beside the sum of its launches' isolated durations (serial replay, event overhead
subtracted).  The ratio says which parts of which queue the other queue slows down, and by how much; the last mark of each queue says
when it really ends.  CT_DUMP=<file> writes the table as JSON."""
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                        # noqa: E402


def main():
    from blobctrl_amd import _lib
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from blobctrl_amd.splat import splat_features
    stride = int(os.environ.get("CT_STRIDE", "6"))
    dev = torch.device("cuda:0")
    ucfg, bcfg = bench.full_configs()
    usd, bsd = bench.synth_weights()
    h = w = 64
    B = int(os.environ.get("CT_BATCH", "1"))
    inp = bench.synth_inputs(h, w, batch=B)
    score = splat_features(**inp["blob"], score_size=(h, w), return_d_score=True, device=str(dev))
    pipe = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device=str(dev), scheduler="ddim")
    pipe(inp["prompt"], inp["fg"], inp["bg"], score, inp["dino"], num_inference_steps=4, latents=inp["latents"])
    P = pipe.plan_for(B, h, w, 77, 768, 4)
    s, side = pipe._streams()
    seg = P.step_active
    lib = _lib.load()

    def reset():
        with torch.cuda.stream(pipe.stream):
            P.step_idx.zero_()
        torch.cuda.synchronize()

    # graph replay of the step (what the edit runs)
    for _ in range(3):
        reset()
        seg.run(s, side, pipe._extra())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        with torch.cuda.stream(pipe.stream):
            P.step_idx.zero_()
        seg.run(s, side, pipe._extra())
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) / 15 * 1e3

    # isolated durations (serial replay, median of 3), event overhead calibrated on a 64-element kernel
    reps = []
    for _ in range(3):
        reset()
        reps.append(seg.run_timed_kernels(s))
        torch.cuda.synchronize()
    n = len(seg.meta)
    tiny = torch.zeros(64, dtype=torch.float16, device=dev)
    ovh = []
    for _ in range(12):
        a_, b_ = C.c_void_p(), C.c_void_p()
        lib.bc_event_create(C.byref(a_)); lib.bc_event_create(C.byref(b_))
        lib.bc_event_record(a_, s)
        lib.bc_silu(tiny.data_ptr(), tiny.data_ptr(), 64, s)
        lib.bc_event_record(b_, s)
        pipe.stream.synchronize()
        ms_ = C.c_float()
        lib.bc_event_elapsed_ms(a_, b_, C.byref(ms_))
        ovh.append(ms_.value)
        lib.bc_event_destroy(a_); lib.bc_event_destroy(b_)
    overhead = sorted(ovh)[len(ovh) // 2]
    real = [m["kind"] not in ("event_record", "event_wait") for m in seg.meta]
    iso = [0.0] * n
    for i in range(n):
        if real[i]:
            d = sorted(r[i][1] + r[i][2] for r in reps)[1]
            nk = 2 if sorted(r[i][2] for r in reps)[1] > 0 else 1
            iso[i] = max(d - overhead, 0.001) + 0.0015 * nk       # + ~1.5 us per dependent kernel boundary (MI355X_MICROARCH: boundary)

    # marks: every `stride`-th real launch of each queue, and each queue's last
    marks = []
    for sid in (0, 1):
        idx = [i for i, m in enumerate(seg.meta) if m["sid"] == sid and real[i]]
        marks += [i for k, i in enumerate(idx) if (k + 1) % stride == 0 or i == idx[-1]]
    marks = sorted(set(marks))
    runs = []
    seg.release()                                             # eager replay (the marks are events between launches)
    for _ in range(7):
        reset()
        t0 = time.perf_counter()
        ms = seg.run_marked(s, side, pipe._extra(), marks)
        runs.append((ms, (time.perf_counter() - t0) * 1e3))
    runs = runs[2:]
    med = [sorted(r[0][k] for r in runs)[len(runs) // 2] for k in range(len(marks))]
    eager_ms = max(med)
    print(f"step: graph replay {graph_ms:.3f} ms; eager concurrent replay with {len(marks)} marks {eager_ms:.3f} ms (host {sorted(r[1] for r in runs)[len(runs) // 2]:.2f} ms); "
          f"event overhead {overhead * 1e3:.1f} us")
    rows = []
    for sid, name in ((0, "UNet"), (1, "BlobNet")):
        prev_i, prev_t = -1, 0.0
        tot_iso = 0.0
        print(f"--- {name} queue: interval (launches) | concurrent ms | isolated ms | ratio | t_end | what")
        for k, i in enumerate(marks):
            if seg.meta[i]["sid"] != sid:
                continue
            span = [j for j in range(prev_i + 1, i + 1) if seg.meta[j]["sid"] == sid and real[j]]
            iso_sum = sum(iso[j] for j in span)
            tot_iso += iso_sum
            conc = med[k] - prev_t
            kinds = {}
            for j in span:
                m = seg.meta[j]
                key = (m.get("variant") or m["kind"]).split("<")[0].replace("_kernel", "") + ":" + m["kind"]
                kinds[key] = kinds.get(key, 0.0) + iso[j]
            what = ", ".join(f"{k_} {v * 1e3:.0f}" for k_, v in sorted(kinds.items(), key=lambda kv: -kv[1])[:4])
            shp = seg.meta[span[-1]].get("shape")
            print(f"  {span[0]:4d}-{span[-1]:4d} | {conc:6.3f} | {iso_sum:6.3f} | {conc / max(iso_sum, 1e-6):5.2f} | {med[k]:6.3f} | {what} | last shape {shp}")
            rows.append(dict(queue=name, first=span[0], last=span[-1], concurrent_ms=round(conc, 4), isolated_ms=round(iso_sum, 4), t_end=round(med[k], 4),
                             kinds={k_: round(v, 4) for k_, v in kinds.items()}, last_shape=shp))
            prev_i, prev_t = i, med[k]
        print(f"  {name}: ends at {prev_t:.3f} ms; isolated sum {tot_iso:.3f} ms")
    if os.environ.get("CT_DUMP"):
        os.makedirs(os.path.dirname(os.environ["CT_DUMP"]) or ".", exist_ok=True)
        with open(os.environ["CT_DUMP"], "w") as f:
            json.dump(dict(graph_ms=graph_ms, eager_ms=eager_ms, overhead_us=overhead * 1e3, stride=stride, intervals=rows), f, indent=1)


if __name__ == "__main__":
    main()
