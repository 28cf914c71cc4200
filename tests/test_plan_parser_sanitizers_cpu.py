"""The `.bcplan` parser under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (SURVEY section 5 "sanitizers").

The file format and every check on it live in blobctrl_amd/csrc/plan_format.h, which makes no HIP call: plan.hip feeds it hipMalloc /
hipMemcpy, the harness tests/c/plan_parse_asan.cpp feeds it calloc / memcpy.  Here the harness is built with
g++ -fsanitize=address,undefined and run over a real tiny plan (compiled without a GPU), every truncation of its header / record
region, a few hundred single-byte corruptions and some hostile length fields: every file must be accepted or rejected with a reason,
never crash, never trip a sanitizer, never yield a pointer outside the arena."""
import os
import random
import struct
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("asan")
    exe = str(d / "plan_parse_asan")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                         os.path.join(REPO, "tests", "c", "plan_parse_asan.cpp"), "-o", exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def plan_bytes(tmp_path_factory):
    """A real plan: the tiny 6-step edit, compiled on the CPU (prologue / active / inactive segments, two stream ids, events)."""
    from blobctrl_amd.pipeline import BlobCtrlEngine
    from tests.common import TINY, tiny_weights
    from tests.gpu_common import tiny_trunk_configs
    usd, bsd = tiny_weights()
    ucfg, bcfg = tiny_trunk_configs()
    eng = BlobCtrlEngine(usd, bsd, ucfg, bcfg, device="cpu", scheduler="ddim", compile_only=True)
    path = str(tmp_path_factory.mktemp("plan") / "t.bcplan")
    eng.compile_plan(path, 1, 8, 8, 7, TINY["ctx"], 4)
    return open(path, "rb").read()


def run(harness, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([harness] + files, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "BAD POINTER" not in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    return lines


def test_parser_accepts_the_real_plan_and_survives_corrupt_copies(harness, plan_bytes, tmp_path):
    raw = plan_bytes
    good = tmp_path / "good.bcplan"
    good.write_bytes(raw)
    first = run(harness, [str(good)])[0]
    assert first.startswith("ok "), first
    nbuf, nseg, nlaunch = (int(x) for x in first.split()[1:4])
    assert nbuf > 50 and nseg >= 3 and nlaunch > 300
    # where the launch records start: the buffer table (with its data blobs) comes first
    magic, version, gsz, nb = struct.unpack("<IIII", raw[:16])
    pos = 16
    for _ in range(nb):
        (ln,) = struct.unpack("<I", raw[pos:pos + 4])
        pos += 4 + ln
        (nbytes,) = struct.unpack("<Q", raw[pos:pos + 8])
        pos += 8
        (has,) = struct.unpack("<I", raw[pos:pos + 4])
        pos += 4 + (nbytes if has else 0)
    rec_start = pos
    assert 16 < rec_start < len(raw)
    files = []

    def emit(b):
        p = tmp_path / f"c{len(files)}.bcplan"
        p.write_bytes(b)
        files.append(str(p))
    rng = random.Random(7)
    # truncations: inside the header / buffer table, inside the record region, and just short of the end
    cuts = list(range(0, 64, 3)) + sorted(rng.sample(range(64, rec_start), 30)) + sorted(rng.sample(range(rec_start, len(raw)), 60)) + [len(raw) - 1]
    for c in cuts:
        emit(raw[:c])
    # single-byte corruptions: the header and buffer-table fields, then the record region (op codes, stream ids, counts, pointers)
    spots = list(range(0, 16)) + sorted(rng.sample(range(16, min(rec_start, 4096)), 60)) + sorted(rng.sample(range(rec_start, len(raw)), 300))
    for s in spots:
        b = bytearray(raw)
        b[s] ^= rng.choice([0x01, 0x80, 0xFF])
        emit(bytes(b))
    # hostile length fields: buffer count, first name length, first buffer size, launch counts
    for off, val in ((12, 0xFFFFFFFF), (12, 1 << 20), (16, 0xFFFFFFF0), (rec_start, 0xFFFFFFFF), (rec_start + 4 + 8 * 16 + 4, 0x7FFFFFFF)):
        b = bytearray(raw)
        b[off:off + 4] = struct.pack("<I", val)
        emit(bytes(b))
    b = bytearray(raw)
    (ln,) = struct.unpack("<I", raw[16:20])
    b[20 + ln:28 + ln] = struct.pack("<Q", 1 << 62)
    emit(bytes(b))
    emit(b"")
    emit(b"BPLN")
    lines = run(harness, files)
    ok = sum(ln.startswith("ok ") for ln in lines)
    rejected = sum(ln.startswith("rejected: ") for ln in lines)
    assert ok + rejected == len(files)
    assert all(ln.startswith("rejected: ") for ln in lines[:len(cuts)]), "a truncated plan was accepted"
    print(f"{len(files)} corrupt copies: {rejected} rejected with a reason, {ok} accepted (a flipped byte inside a weight blob or a scalar argument is still a valid file)")
