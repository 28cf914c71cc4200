#!/usr/bin/env python3
"""Reproducer probe for the round-1 observation that `hipStreamEndCapture` crashed in the runtime on a step graph with ~100
fork / join pairs over five streams (DESIGN 3.4 "Auxiliary streams inside a block").  Builds a segment of PAIRS fork/join pairs over
STREAMS streams with a trivial kernel on every branch (through the plan runtime's signal / wait ops), captures it, replays it and
checks the result.  Each configuration runs in its own child process (a runtime crash must not take the sweep down).
Usage: python tools/capture_repro.py            (sweep)      |   python tools/capture_repro.py PAIRS STREAMS   (one case)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one(pairs, nstreams, unjoined=""):
    import torch
    from blobctrl_amd.launch import Recorder
    dev = torch.device("cuda:0")
    rec = Recorder(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    seg = rec.begin("repro")
    x = torch.full((nstreams, 256), 0.5, dtype=torch.float16, device=dev)
    y = torch.zeros_like(x)
    rec.keep.append((x, y))
    for i in range(pairs):
        sid = 1 + i % (nstreams - 1)
        fork = rec.new_event()
        rec.sid = 0
        rec._op("bc_silu", (x[0], y[0], 256), "silu")
        rec.signal(fork)
        rec.sid = sid
        rec.wait(fork)
        rec._op("bc_silu", (x[sid], y[sid], 256), "silu")
        join = rec.new_event()
        rec.signal(join)
        rec.sid = 0
        rec.wait(join)
    if unjoined == "wait":   # a forked, properly joined stream whose LAST recorded operation is a wait on another stream's event
        fork = rec.new_event()
        rec.signal(fork)
        rec.sid = 1
        rec.wait(fork)
        rec._op("bc_silu", (x[1], y[1], 256), "silu")
        j1 = rec.new_event()
        rec.signal(j1)
        rec.sid = 2
        rec.wait(fork)
        rec._op("bc_silu", (x[2], y[2], 256), "silu")
        e2 = rec.new_event()
        rec.signal(e2)
        rec.sid = 1
        rec.wait(e2)             # dangling: nothing follows on stream 1, nobody waits for stream 1 after this
        rec.sid = 0
        rec.wait(j1)
        rec.wait(e2)
    elif unjoined:       # a forked stream whose last launch nobody waits for: hipStreamEndCapture should return an error ...
        fork = rec.new_event()
        rec.signal(fork)
        rec.sid = 1
        rec.wait(fork)
        rec._op("bc_silu", (x[1], y[1], 256), "silu")
        rec.sid = 0
    s = [st.cuda_stream for st in streams]
    seg.capture(s[0], s[1], s[2:])
    for _ in range(3):
        seg.run(s[0], s[1], s[2:])
    torch.cuda.synchronize()
    ref = torch.nn.functional.silu(torch.tensor(0.5))
    ok = bool(torch.allclose(y.float().cpu(), torch.full((nstreams, 256), float(ref)), atol=1e-3))
    print(f"pairs={pairs} streams={nstreams}: captured + replayed, result {'ok' if ok else 'WRONG'}", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        sys.exit(one(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else ""))
    for pairs, ns, uj in ((8, 3, ""), (100, 5, ""), (300, 5, ""), (100, 8, ""), (8, 3, "kernel"), (8, 3, "wait")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(pairs), str(ns)] + ([uj] if uj else []),
                           capture_output=True, text=True, timeout=300)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l][-2:]
        print(f"[{pairs} pairs / {ns} streams{' + unjoined tail: ' + uj if uj else ''}] rc={r.returncode}: " + " | ".join(tail), flush=True)
