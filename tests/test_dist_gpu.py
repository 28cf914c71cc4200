"""The multi-rank path on the GPU box: `python bench.py --gpus 2` started as a fresh child process.  A 1-GPU box
makes the launcher fall back to gloo with both ranks on the one device (RCCL refuses two ranks per device), so this exercises the
launcher, the rendezvous, the weight-arena broadcast into a second process, request sharding and the max-over-ranks timing on real
HIP kernels - everything except RCCL's own transport, which needs the driver's multi-GPU tier."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_identical_weights_and_both_run_edits():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--denoise-steps", "4",
                        "--no-cpu-baseline", "--no-roofline", "--no-e2e", "--no-configs"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and len(cfg["ranks"]) == 2 and cfg["nranks_seen"] == 2
    assert sorted(x["rank"] for x in cfg["ranks"]) == [0, 1]
    assert cfg["dist_backend"] in ("gloo", "nccl")
    # rank 1 received exactly the arenas rank 0 packed
    assert len(cfg["weights_sha16"]) == 2 and cfg["weights_sha16"][0] == cfg["weights_sha16"][1] and len(cfg["weights_sha16"][0]) == 16
    assert line["value"] > 0 and cfg["edits_per_rank"] == 1
    if cfg["dist_backend"] == "nccl":
        assert cfg["rccl_version"]


def test_c4_per_request_batches_replay_one_captured_graph():
    """BASELINE configs[3] gives every rank 8 per-request batches of 8; the first hardware run on an 8-GPU node must
    measure kernels, not captures.  A small stand-in on this box (6 requests, batches of 2, 2 denoise steps, one rank): the engine records
    ONE plan and captures ONE whole-edit graph in the warm-up batch, and every timed batch is a cache hit on both."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--requests", "6", "--batch", "2", "--warmup", "1", "--denoise-steps", "2",
                        "--no-cpu-baseline", "--no-roofline", "--no-e2e", "--no-configs", "--no-calibration"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    pc = line["config"]["plan_cache"]
    assert pc["plans_recorded"] == 1 and pc["loop_graph_captures"] == 1 and pc["loop_graph_hits"] == 3, pc
    assert line["config"]["requests_total"] == 6 and line["config"]["per_request_batches"] and line["value"] > 0
    assert line["config"]["plan"]["options_non_default"] == {} and line["config"]["plan"]["variants"]
