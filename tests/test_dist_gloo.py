"""N>1 path on CPU: two gloo ranks.  Checks the shard arithmetic, the single weight broadcast and
the counter gather used by bench.py (the GPU run uses the same code over RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}


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


@pytest.mark.timeout(300)
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` must itself bring up 2 ranks: the parent spawns the children
    before touching any GPU, rank 0's JSON line reports n_gpus = 2, the weight broadcast and per-rank counters ran.
    --dry-run skips the timed GPU step (there is no GPU here); on a GPU box the same launcher path runs the real step."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--dry-run", "--steps", "3", "--workload", "336x256"], env=env, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True
    assert d["weights_broadcast"]["bytes"] > 0
    assert d["per_rank_images"] == [256.0 * 3, 256.0 * 3]
    assert d["config"]["batch_per_gpu"] == 256
    assert "also_336x256" not in d             # the legs ride on the default 1024 line only


def _check_dist_legs(d, n):
    """The legs EVERY rank of a group runs behind the main 1024 line: BASELINE configs[3] (256 images of
    336 x 336 per rank) and the ragged main_batched chain, each with the job rate, every rank's rate, the rate against N x the
    mean rank, and the AND over ranks of its bit-identity check."""
    assert "driver" in d["scaling_curve"] and d["rccl_ranks_seen"] == list(range(n))
    for key in ("also_336x256", "also_main_batched_ragged"):
        leg = d[key]
        assert leg["n_gpus"] == n and len(leg["per_rank_images_per_s"]) == n and all(v > 0 for v in leg["per_rank_images_per_s"])
        assert leg["bit_identical_to_serial"] is True and 0.0 < leg["scaling_efficiency_vs_rank_mean"] <= 1.0 + 1e-6
        mean = sum(leg["per_rank_images_per_s"]) / n
        # (the fields are rounded to 0.1 image/s: compare with that slack)
        assert abs(leg["scaling_efficiency_vs_rank_mean"] - leg["value"] / (n * mean)) < 1e-3 + 0.2 / mean


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 8])
def test_bench_default_workload_carries_the_dist_legs(n):
    """The command the driver runs for the scaling curve is `bench.py --gpus N` with the DEFAULT workload: its line must carry
    configs[3] and the ragged main_batched leg from every rank, not the 1024 x 1024 workload alone."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "gloo", "--dry-run",
                        "--steps", "2"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == n and d["config"]["workload"] == "1024"
    _check_dist_legs(d, n)
    # --legs narrows them like every other leg
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--dry-run",
                        "--steps", "2", "--legs", "336"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "also_336x256" in d and "also_main_batched_ragged" not in d


def test_bench_exits_nonzero_when_a_rank_is_missing_from_the_gather(monkeypatch):
    """`rccl_ranks_seen` is asserted, not only reported: a group whose gather does not return ranks 0 .. N-1 ends the run."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.argv = ['bench.py', '--gpus', '1', '--force-dist', '--dry-run', '--dist-backend', 'gloo']\n"
            "import bench\nfrom attwarp_amd import dist as D\n"
            "orig = D.all_gather_counters\n"
            "D.all_gather_counters = lambda v: {k: [7.0] for k in v} if 'rank' in v else orig(v)\n"
            "bench.main()") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "gathered ranks [7]" in (r.stderr + r.stdout), r.stderr[-1500:]


def test_bench_refuses_world_size_mismatch():
    """A stale WORLD_SIZE must not silently produce a 1-GPU number labelled as N."""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.timeout(400)
def test_bench_self_launches_eight_ranks():
    """The shape the driver's scaling run has: `bench.py --gpus 8` brings up 8 ranks that all see each other
    (all_gather of the ranks), each pinned to its own share of the host cores before torch is imported."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo",
                        "--dry-run", "--steps", "2", "--workload", "336x256"], env=_clean_env(), capture_output=True,
                       text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == list(range(8))
    assert d["per_rank_images"] == [512.0] * 8
    assert d["config"]["batch_per_gpu"] == 256
    assert d["cpu_affinity_rank0"] is None or "cores" in d["cpu_affinity_rank0"]


@pytest.mark.timeout(200)
def test_bench_launcher_fails_fast_when_a_rank_dies():
    """Rank 3 exits at start-up: the parent must notice (it polls every child, not only rank 0), kill the ranks that
    are now stuck in the rendezvous and return non-zero within seconds -- not after torch's collective timeout."""
    import subprocess
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dist-backend", "gloo",
                        "--dry-run", "--steps", "2"], env=dict(_clean_env(), ATTWARP_BENCH_FAIL_RANK="3"),
                       capture_output=True, text=True, timeout=180)
    dt = time.monotonic() - t0
    assert r.returncode != 0
    assert "rank(s) failed" in r.stderr and "(3, 3)" in r.stderr, r.stderr[-1500:]
    assert dt < 30.0, dt
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_pin_to_local_cores_is_harmless_without_a_gpu():
    """No KFD topology here: the even split of the allowed cores is used; the process keeps a non-empty affinity."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import bench; before = os.sched_getaffinity(0); "
            "d = bench.pin_to_local_cores(1, 4); after = os.sched_getaffinity(0); "
            "assert after and after <= before, (before, after); print(d)") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "cores" in r.stdout


@pytest.mark.timeout(400)
def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- RANK is set, so bench.py is simply one of the ranks (no self-launch);
    rank 0 prints the one JSON line with the dist legs."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dist-backend", "gloo", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, (r.stderr[-2000:], r.stdout[-500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1]
    _check_dist_legs(d, 2)
