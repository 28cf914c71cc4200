"""World-size-2 gloo test of the multi-GPU path: rank 0 packs the weights, both ranks end with bit-identical replicas,
requests are sharded round-robin with no data-path collective, results are gathered."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blobctrl_amd import dist as bdist, synth
from blobctrl_amd.weights import PackedTrunk
from tests.common import TINY


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = bdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    built = []

    def build():
        built.append(rank)
        sh = synth.trunk_param_shapes(5, TINY["boc"], 2, TINY["ctx"], 4, blobnet=False)
        return PackedTrunk(synth.synth_state_dict(sh, 7), torch.device("cpu"), TINY["boc"])

    pw = bdist.broadcast_packed(build, torch.device("cpu"))
    assert built == ([0] if rank == 0 else []), "only rank 0 may pack the weights"
    mine = bdist.shard_requests(5, rank, world)
    local = [torch.full((2,), float(i)) for i in mine]
    gathered = bdist.gather_results(local, dst=0)
    tmax = bdist.barrier_max_seconds(1.0 + rank)
    torch.save(dict(h=pw.h_arena.clone(), f=pw.f_arena.clone(), keys=sorted(pw.h.keys()), mine=mine, tmax=tmax,
                    temb=pw.temb_total, gathered=gathered), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_request_sharding_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    b = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    assert torch.equal(a["h"], b["h"]) and torch.equal(a["f"], b["f"]) and a["keys"] == b["keys"]
    assert a["h"].abs().sum() > 0 and a["temb"] == b["temb"] > 0
    assert a["mine"] == [0, 2, 4] and b["mine"] == [1, 3]
    assert a["tmax"] == b["tmax"] == 2.0
    flat = sorted(float(t[0]) for part in a["gathered"] for t in part)
    assert flat == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_single_process_paths_are_noops():
    assert bdist.shard_requests(3, 0, 1) == [0, 1, 2]
    assert bdist.barrier_max_seconds(0.5) == 0.5


def test_config_c4_partition_64_requests_over_8_ranks():
    """BASELINE configs[3]: 64 edits, 8 per GPU, no request on two ranks, none dropped."""
    parts = [bdist.shard_requests(64, r, 8) for r in range(8)]
    assert all(len(p) == 8 for p in parts)
    assert sorted(i for p in parts for i in p) == list(range(64))
