"""Critical path of the two-stream step from its own launch list (GPU).

Every launch of the active step is timed alone (HIP events around each kernel, serial replay: `bc_plan_run_timed_kernels`), then the
two queues are replayed ON PAPER with those durations and the plan's event edges (fork, one event per BlobNet residual): the result is
what the step would cost if the two streams did not slow each other down - and, more to the point, WHERE the UNet queue sits waiting
for a BlobNet residual and which queue ends the step.  Prints the per-stream busy time, every stall of more than 5 us, and the
makespan of the paper schedule beside the measured step."""
import os
import sys
import time
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench                                                        # noqa: E402


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
    seg = P.step_active

    def run():
        with torch.cuda.stream(pipe.stream):
            P.step_idx.zero_()
        seg.run(s, side)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        run()
    torch.cuda.synchronize()
    measured = (time.perf_counter() - t0) / 15 * 1e3

    # serial per-launch durations (median of 3 timed replays)
    reps = []
    for _ in range(3):
        with torch.cuda.stream(pipe.stream):
            P.step_idx.zero_()
        reps.append(seg.run_timed_kernels(s))
        torch.cuda.synchronize()
    n = len(seg.meta)
    dur = [sorted(r[i][1] + r[i][2] for r in reps)[1] for i in range(n)]
    if os.environ.get("CP_DUMP"):
        # per-launch table (index, stream, kind, variant, shape, main-kernel us, reducer us) for offline analysis
        import json
        rows = [dict(i=i, sid=m["sid"], kind=m["kind"], variant=m.get("variant", ""), shape=m.get("shape"), flops=m.get("flops", 0),
                     us=round(sorted(r[i][1] for r in reps)[1] * 1e3, 2), red_us=round(sorted(r[i][2] for r in reps)[1] * 1e3, 2))
                for i, m in enumerate(seg.meta)]
        os.makedirs(os.path.dirname(os.environ["CP_DUMP"]) or ".", exist_ok=True)
        with open(os.environ["CP_DUMP"], "w") as f:
            json.dump(dict(measured_ms=measured, launches=rows), f)

    ev_of = {i: m["ev"] for i, m in enumerate(seg.meta) if "ev" in m}
    clock = {0: 0.0, 1: 0.0, 2: 0.0}
    busy = {0: 0.0, 1: 0.0, 2: 0.0}
    signalled = {}
    stalls, slack = [], []
    last_kind = {0: "", 1: "", 2: ""}
    for i, m in enumerate(seg.meta):
        sid = m["sid"]
        if m["kind"] == "event_record":
            signalled[ev_of[i]] = clock[sid]
            continue
        if m["kind"] == "event_wait":
            t = signalled.get(ev_of[i])
            if t is None:
                raise SystemExit(f"wait on an event not yet recorded at launch {i}")
            if t > clock[sid] + 0.005:
                stalls.append((i, sid, clock[sid], t - clock[sid], last_kind[sid]))
            if sid == 0:
                slack.append((i, clock[sid], t))
            clock[sid] = max(clock[sid], t)
            continue
        clock[sid] += dur[i]
        busy[sid] += dur[i]
        last_kind[sid] = f"{m['kind']} {m.get('variant', '')} {m.get('shape')}"
    print(f"measured step (graph replay, two streams): {measured:.3f} ms")
    print(f"serial kernel time: stream 0 (UNet) {busy[0]:.3f} ms, stream 1 (BlobNet) {busy[1]:.3f} ms, sum {busy[0] + busy[1]:.3f} ms")
    print(f"paper schedule without interference: UNet queue ends at {clock[0]:.3f} ms, BlobNet queue at {clock[1]:.3f} ms")
    tot = 0.0
    for i, sid, at, d, after in stalls:
        tot += d
        print(f"  stream {sid} stalls {d * 1e3:7.1f} us at t = {at:.3f} ms (launch {i}, after {after})")
    print(f"total stall on the paper schedule: {tot:.3f} ms")
    print("UNet residual waits (UNet clock when it needs the residual, BlobNet clock when it was produced):")
    for i, at, t in slack:
        print(f"  launch {i}: UNet at {at:.3f} ms, residual ready at {t:.3f} ms, slack {at - t:+.3f} ms")
    # where each queue is when the other one finishes its phases
    unet_cum, t = [], 0.0
    for i, m in enumerate(seg.meta):
        if m["sid"] == 0 and m["kind"] not in ("event_record", "event_wait"):
            t += dur[i]
            unet_cum.append((i, t))
    for sid, name in ((0, "UNet"), (1, "BlobNet")):
        kinds = {}
        for i, m in enumerate(seg.meta):
            if m["sid"] == sid and m["kind"] not in ("event_record", "event_wait"):
                k = (m.get("variant") or m["kind"]).split("<")[0] + ("/" + m["kind"] if m["kind"] not in (m.get("variant") or "") else "")
                a = kinds.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += dur[i]
        print(f"{name} queue by kernel / op kind (launches, ms):")
        for k, (n_, ms) in sorted(kinds.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"    {k:44s} {n_:4d} {ms:7.3f}")
    first_wait = next((i for i, m in enumerate(seg.meta) if m["sid"] == 0 and m["kind"] == "event_wait" and i > 2), None)
    if first_wait is not None:
        before = sum(dur[i] for i, m in enumerate(seg.meta) if m["sid"] == 0 and i < first_wait and m["kind"] not in ("event_record", "event_wait"))
        print(f"UNet work recorded before its first residual wait: {before:.3f} ms of {busy[0]:.3f}")


if __name__ == "__main__":
    main()
