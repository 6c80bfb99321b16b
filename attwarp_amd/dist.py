"""Multi-GPU harness: one process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).

The path shards embarrassingly: every image (with its attention stack) is warped independently, so
ranks take contiguous blocks of the batch and never exchange image data.  The single collective is
one broadcast of the flattened MarginalNet weights (11 MB at hidden=256) from rank 0 at start-up
(SURVEY 8e); an optional all_gather moves a few counters for reporting.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
import torch.distributed as dist


def _grouped() -> bool:
    """True when the collectives below have a process group to run on (any world size, including a one-rank group)."""
    return dist.is_available() and dist.is_initialized()


def init(backend: str | None = None, device_index: int | None = None, use_gpu: bool = True,
         force_group: bool = False) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank).  Single-process runs need no environment.
    ``device_index`` overrides the GPU (default: LOCAL_RANK) -- only used to smoke-test the N>1 code
    path on a one-GPU box with the gloo backend.  ``use_gpu=False`` (bench.py --dry-run, CPU tests) never touches
    the GPU and defaults to gloo.
    ``force_group=True`` builds the process group even for WORLD_SIZE == 1 (rendezvous on 127.0.0.1 when MASTER_* are
    absent): every helper below then runs its real collective -- on a one-GPU box that is a one-rank RCCL communicator,
    which loads librccl, creates the communicator on the device and executes broadcast / all_gather / all_reduce /
    barrier exactly as a rank of N would (``bench.py --gpus 1 --force-dist``, tests/test_gpu_parity.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev_idx = local if device_index is None else device_index
    have_gpu = use_gpu and torch.cuda.is_available()
    if have_gpu:
        torch.cuda.set_device(dev_idx)
    if (world > 1 or force_group) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if have_gpu else "gloo"
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        kw = {}
        if backend == "nccl":                 # bind the communicator to this rank's GPU up front (no lazy device guess)
            kw["device_id"] = torch.device("cuda", dev_idx)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)   # "nccl" is RCCL on ROCm
    return rank, world, local


def shutdown():
    """Destroy the process group if there is one (a forced one-rank group included)."""
    if _grouped():
        dist.destroy_process_group()


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_module_weights(module: torch.nn.Module, src: int = 0) -> int:
    """Broadcast every parameter and buffer of ``module`` from ``src`` as ONE flat message.
    Returns the number of bytes moved (0 without a process group)."""
    if not _grouped():
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return 0
    dev = tensors[0].device
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    if dist.get_backend() == "gloo":          # CPU smoke tests: stage through host memory
        host = flat.cpu()
        dist.broadcast(host, src=src)
        flat = host.to(dev)
    else:
        flat = flat.to(dev)
        dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n
    return flat.numel() * 4


def all_gather_counters(values: Dict[str, float]) -> Dict[str, list]:
    """Gather a few per-rank scalars (image counts, timings) on every rank."""
    keys = sorted(values)
    if not _grouped():
        return {k: [float(values[k])] for k in keys}
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return {k: [float(o[i]) for o in out] for i, k in enumerate(keys)}


def barrier():
    if _grouped():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float) -> float:
    if not _grouped():
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
