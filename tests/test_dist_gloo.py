"""N>1 path on CPU: two gloo ranks.  Checks the shard arithmetic, the single weight broadcast and
the counter gather used by bench.py (the GPU run uses the same code over RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from attwarp_amd import dist as d
    from attwarp_amd.model import MarginalNet
    r, w, _ = d.init(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)               # different random init per rank
    net = MarginalNet(8, 12, hidden=4)
    nbytes = d.broadcast_module_weights(net, src=0)
    checksum = float(sum(p.double().sum() for p in net.parameters()))
    lo, hi = d.shard_range(11, rank, world)
    gathered = d.all_gather_counters({"images": hi - lo, "ms": 1.5 + rank})
    mx = d.max_over_ranks(10.0 * (rank + 1))
    d.barrier()
    q.put((rank, nbytes, checksum, (lo, hi), gathered, mx))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, nb0, c0, s0, g0, m0), (r1, nb1, c1, s1, g1, m1) = res
    assert nb0 == nb1 > 0
    assert c0 == c1                                # rank 1 received rank 0's weights exactly
    assert s0 == (0, 6) and s1 == (6, 11)
    assert g0 == g1 == {"images": [6.0, 5.0], "ms": [1.5, 2.5]}
    assert m0 == m1 == 20.0
