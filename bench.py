#!/usr/bin/env python3
"""bench.py -- warped images/sec of the AttWarp hot path on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path over one batch of synthetic inputs already resident in HBM:

    attention rows [T=20, B, 32 heads, kv=640] float32
      -> A1+A2 aggregation -> 24x24 map -> A6 marginals -> A8+A9+A11 PDF -> CDF -> inverse maps
      -> A12 bilinear resample of images [B, S, S, 3] float32 (HWC)  -> warped [B, S, S, 3]

Default workload = BASELINE configs[2] (B=256 per GPU, S=1024: the configuration the 70 %-of-roofline
target is quoted on; it fits one GPU: 3.2 GB in + 3.2 GB out).  ``--workload 336`` runs configs[1]
(B=64, S=336); the 336 result is also attached to the default line under "also".  With N GPUs every
rank processes its own B images (weak scaling, the path shards by image, no data-path collective);
the only collective is the start-up RCCL broadcast of MarginalNet weights, untimed.

Output: ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     dominant kernel (remap_rows_kernel): algorithmic bytes (2*S*S*3*4 per image)
               / its mean launch duration measured live with HIP events inside the timed region,
               against the 8 TB/s HBM peak.
  cpu_baseline the plain-C restatement of the same path (oracle/warp_ref.c, "port") timed on this
               box's host on a bounded sample, single thread.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
T_STEPS, HEADS, KV, NTOK = 20, 32, 640, 576
WORKLOADS = {"1024": (256, 1024), "336": (64, 336)}


def make_inputs(B: int, S: int, dev, seed: int):
    """Synthetic inputs of SURVEY 8d: uniform [0,1) float32 HWC images; attention rows = softmax of a
    640-wide random-normal row, image tokens at starts = 35 + (b mod 8)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    img = torch.rand((B, S, S, 3), device=dev, generator=g)
    rows = torch.empty((T_STEPS, B, HEADS, KV), device=dev)
    for t in range(T_STEPS):                       # chunked: keeps the temporary small
        rows[t] = torch.softmax(torch.randn((B, HEADS, KV), device=dev, generator=g), dim=-1)
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    return img, rows, starts


class Step:
    """The hot-path step on static buffers, with HIP events around the dominant kernel."""

    def __init__(self, B, S, dev, seed):
        from attwarp_amd import attention_extraction as ae, checkpoint_utils as cu, pipeline
        self.ae, self.cu, self.pipeline = ae, cu, pipeline
        self.B, self.S = B, S
        self.img, self.rows, self.starts = make_inputs(B, S, dev, seed)
        self.out = torch.empty_like(self.img)
        self.starts_tiled = self.starts.repeat(T_STEPS)
        self.events = []

    def __call__(self, record: bool = False):
        # same three launches as attwarp_amd.pipeline.warp_from_attention_stack, with HIP events between them
        # (torch's current stream == the stream the kernels are launched on)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if record:
            ev[0].record()
        steps = self.pipeline.attention_step_maps(self.rows, self.starts, NTOK, self.starts_tiled)
        if record:
            ev[1].record()
        mx, my = self.pipeline.axis_maps_from_attention_steps(steps, (self.S, self.S))
        if record:
            ev[2].record()
        self.cu.remap_separable(self.img, mx, my, channels_last=True, out=self.out)
        if record:
            ev[3].record()
            self.events.append(ev)
        return self.out

    def stage_ms(self):
        """Mean duration of the three kernels of a step: (attention reduce, maps, resample)."""
        return [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in self.events])) for i in range(3)]

    def remap_ms(self):
        return [e[2].elapsed_time(e[3]) for e in self.events]


def run_workload(name, steps, warmup, dist_mod, dev, rank):
    B, S = WORKLOADS[name]
    step = Step(B, S, dev, seed=1234 + rank)
    for _ in range(warmup):
        step()
    dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(record=True)
    torch.cuda.synchronize()
    dist_mod.barrier()
    wall = time.perf_counter() - t0
    wall = dist_mod.max_over_ranks(wall)
    remap = step.remap_ms()
    return step, wall, remap


def cpu_baseline(step: "Step", budget_s: float = 12.0):
    """Time the plain-C port of the same path (oracle/warp_ref.c) on this host: one thread, a bounded
    sample of the batch (whole images through attention reduce -> ... -> remap)."""
    from oracle import c_oracle
    from attwarp_amd import _tables
    c_oracle.load()
    S = step.S
    inv = _tables._right_inverse_inv_host(24, S, 1e-8)
    n_max = min(step.B, 64)
    img = step.img[:n_max].cpu().numpy()
    rows = step.rows[:, :n_max].cpu().numpy()
    starts = step.starts[:n_max].cpu().numpy()
    out0 = c_oracle.warp_from_attention_stack(img[0], rows[:, 0], starts[0], inv, inv)     # warm caches
    t0 = time.perf_counter()
    n = 0
    while True:                                   # cycle over the sample until the time budget is spent
        c_oracle.warp_from_attention_stack(img[n % n_max], rows[:, n % n_max], starts[n % n_max], inv, inv)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 4:
            break
    dt = time.perf_counter() - t0
    # the sample doubles as a parity check of what was just benchmarked
    err = float(np.abs(step.out[0].cpu().numpy() - out0).max())
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{n} image passes over the first {n_max} of the {step.B} {S}x{S} images through "
                      f"oracle/warp_ref.c (attention reduce .. remap), 1 thread of {os.cpu_count()} host cores, "
                      f"{dt:.1f} s",
            "max_abs_diff_vs_gpu_image0": err}


def load_pmc_traffic(workload: str):
    """HBM bytes per launch of the remap kernel from the committed rocprofv3 --pmc summary
    (profiles/round1_pmc.json, collected with this same command).  None if absent."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        return d.get(workload, {}).get("remap_rows_kernel_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1024")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary 336x336 measurement")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="force the GPU index (smoke-testing the N>1 path on a one-GPU box with --dist-backend gloo)")
    args = ap.parse_args()

    from attwarp_amd import dist as D, _lib
    rank, world, local = D.init(args.dist_backend, args.device)
    if args.device is not None:
        local = args.device
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    _lib.load()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # start-up collective of the multi-GPU path: one RCCL broadcast of MarginalNet weights (untimed)
    bcast = None
    if world > 1:
        from attwarp_amd.model import MarginalNet
        net = MarginalNet(1024, 4096, 256).to(dev)
        torch.cuda.synchronize(); D.barrier()
        t0 = time.perf_counter()
        nbytes = D.broadcast_module_weights(net, src=0)
        torch.cuda.synchronize()
        bcast = {"bytes": nbytes, "ms": round((time.perf_counter() - t0) * 1e3, 3)}
        del net

    step, wall, remap = run_workload(args.workload, args.steps, args.warmup, D, dev, rank)
    B, S = WORKLOADS[args.workload]
    ms_per_step = wall / args.steps * 1e3
    value = world * B * args.steps / wall
    remap_ms = float(np.mean(remap))
    alg_bytes = 2.0 * S * S * 3 * 4 * B
    achieved = alg_bytes / (remap_ms * 1e-3) / 1e9

    result = {
        "metric": "warped images/sec",
        "value": round(value, 1),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"batch-{B} {S}x{S}x3 float32 HWC images per GPU + attention rows "
                               f"[T={T_STEPS},B,{HEADS},{KV}] float32 -> reduce -> 24x24 -> marginals -> CDF -> "
                               f"inverse maps -> bilinear warp (BASELINE configs[{2 if S == 1024 else 1}])",
                   "batch_per_gpu": B, "image_size": S, "layout": "HWC", "global_batch": world * B,
                   "sharding": "contiguous image blocks per rank, no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": "remap_rows_kernel", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": load_pmc_traffic(args.workload),
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_mean": round(remap_ms, 4),
                     "kernel_ms_min": round(float(np.min(remap)), 4), "launches_timed": len(remap)},
    }
    st_ms = step.stage_ms()
    result["stages_ms"] = {"attn_reduce_step_kernel": round(st_ms[0], 4), "axis_maps_from_steps_kernel": round(st_ms[1], 4),
                           "remap_rows_kernel": round(st_ms[2], 4)}
    if bcast:
        result["weights_broadcast"] = bcast

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(step)
    del step
    torch.cuda.empty_cache()

    if args.workload == "1024" and not args.no_also:
        step2, wall2, remap2 = run_workload("336", max(args.steps, 50), args.warmup, D, dev, rank)
        B2, S2 = WORKLOADS["336"]
        n2 = max(args.steps, 50)
        rm2 = float(np.mean(remap2))
        ach2 = 2.0 * S2 * S2 * 3 * 4 * B2 / (rm2 * 1e-3) / 1e9
        result["also"] = {"workload": f"batch-{B2} {S2}x{S2} per GPU (BASELINE configs[1]), eager launches (a HIP-graph replay of the "
                                      f"same 3 kernels measured identical: the step is GPU-bound)",
                          "value": round(world * B2 * n2 / wall2, 1), "unit": "images/s",
                          "ms_per_step": round(wall2 / n2 * 1e3, 4),
                          "roofline": {"achieved": round(ach2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(ach2 / HBM_PEAK_GBS, 4), "kernel_ms_mean": round(rm2, 4)}}
        del step2

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
