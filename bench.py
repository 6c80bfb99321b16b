#!/usr/bin/env python3
"""bench.py -- warped images/sec of the AttWarp hot path on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path over one batch of synthetic inputs already resident in HBM:

    attention rows [T=20, B, 32 heads, kv=640] float32
      -> A1+A2 aggregation -> 24x24 map -> A6 marginals -> A8+A9+A11 PDF -> CDF -> inverse maps
      -> A12 bilinear resample of images [B, S, S, 3] float32  -> warped [B, S, S, 3]

Workloads (``--workload``):
  1024     BASELINE configs[2]: B=256 per GPU, S=1024 (default; the configuration the 70 %-of-roofline target is
           quoted on; 3.2 GB in + 3.2 GB out per GPU)
  336      BASELINE configs[1]: B=64, S=336
  336x256  BASELINE configs[3]: B=256 per GPU, S=336 (2048 images over 8 ranks)
Arithmetic (``--mode``): ``cv2`` (default) = what the reference's cv2.remap(INTER_LINEAR) call computes (1/32-px
quantised coordinates, 4 table weights); ``exact`` = unquantised bilinear (= F.grid_sample).  Both run on the same
staged kernel; the default line measures cv2 and attaches the exact-mode, CHW-layout and 336 measurements under
"also_exact", "also_chw", "also".

``--gpus N`` (N > 1) is self-launching: when RANK is not in the environment the parent -- before it touches the GPU --
starts N child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), one per GPU,
relays rank 0's JSON line and exits non-zero if any rank failed.  Launched under ``torch.distributed.run`` (RANK set)
it simply is one of the ranks.  Every rank processes its own B images (weak scaling, the path shards by image, no
data-path collective); the only collective is the start-up RCCL broadcast of MarginalNet weights, untimed.

Output: ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     dominant kernel (remap_rows_kernel): algorithmic bytes (2*S*S*3*4 per image)
               / its mean launch duration measured live with HIP events inside the timed region,
               against the 8 TB/s HBM peak.
  cpu_baseline the plain-C restatement of the same path (oracle/warp_ref.c, "port") timed on this
               box's host on a bounded sample: one thread, and one process per host core.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
T_STEPS, HEADS, KV, NTOK = 20, 32, 640, 576
WORKLOADS = {"1024": (256, 1024, 2), "336": (64, 336, 1), "336x256": (256, 336, 3)}   # B per GPU, S, BASELINE config
PREWARM_STEPS = 25             # untimed device pre-conditioning before the W warm-up steps (see main())


# ----------------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N  ->  N ranks.  Runs before torch is imported; never touches the GPU.
# ----------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, ATTWARP_BENCH_CHILD="1")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    try:
        line0, _ = procs[0].communicate(timeout=float(os.environ.get("ATTWARP_BENCH_TIMEOUT_S", "3000")))
        rcs = [procs[0].returncode] + [p.wait(timeout=120) for p in procs[1:]]
    except subprocess.TimeoutExpired:
        print("[bench] ranks did not finish in time; terminating them", file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.kill()                         # the exact PIDs this parent started
        return 1
    if any(rcs):
        print(f"[bench] rank exit codes {rcs}: at least one rank failed", file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.kill()
        return 1
    lines = [l for l in line0.decode().splitlines() if l.startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle/warp_ref.c through oracle/c_oracle.py -- the checker, timed beside the product)
# ----------------------------------------------------------------------------------------------------------------
def _cpu_inputs(S: int, n: int, seed: int):
    """Host-side synthetic inputs of the same distributions as make_inputs()."""
    import numpy as np
    rng = np.random.default_rng(seed)
    img = rng.random((n, S, S, 3), dtype=np.float32)
    lg = rng.standard_normal((T_STEPS, n, HEADS, KV)).astype(np.float32)
    e = np.exp(lg - lg.max(-1, keepdims=True))
    rows = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    starts = (35 + np.arange(n) % 8).astype(np.int32)
    return img, rows, starts


def _cpu_loop(img, rows, starts, S, mode, budget_s):
    from oracle import c_oracle, warp_oracle as O
    inv = O.right_inverse_core(24, S)              # the oracle's own table (not the product's)
    n_max = img.shape[0]
    out0 = c_oracle.warp_from_attention_stack(img[0], rows[:, 0], starts[0], inv, inv, mode=mode)   # warm caches
    t0 = time.perf_counter()
    n = 0
    while True:                                    # cycle over the sample until the time budget is spent
        c_oracle.warp_from_attention_stack(img[n % n_max], rows[:, n % n_max], starts[n % n_max], inv, inv, mode=mode)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 2:
            break
    return n, time.perf_counter() - t0, out0


def _cpu_worker(idx, S, mode, budget_s, start_evt, q):
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[idx % len(os.sched_getaffinity(0))]})
    except Exception:
        pass
    img, rows, starts = _cpu_inputs(S, 2, 9000 + idx)
    start_evt.wait()
    n, dt, _ = _cpu_loop(img, rows, starts, S, mode, budget_s)
    q.put((n, dt))


def cpu_baseline_all_cores(S: int, mode: str, budget_s: float = 10.0):
    """One process per host core over disjoint synthetic image shards.  MUST run before this process
    initialises the GPU: the workers are forked."""
    import multiprocessing as mp
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ctx = mp.get_context("fork")
    q, evt = ctx.Queue(), ctx.Event()
    procs = [ctx.Process(target=_cpu_worker, args=(i, S, mode, budget_s, evt, q)) for i in range(ncores)]
    for p in procs:
        p.start()
    time.sleep(min(20.0, 1.0 + 0.02 * ncores + (3.0 if S >= 1024 else 0.5)))   # let every worker build its inputs
    t0 = time.perf_counter()
    evt.set()
    res = [q.get(timeout=budget_s * 6 + 120) for _ in procs]
    wall = time.perf_counter() - t0
    for p in procs:
        p.join(30)
    total = sum(n for n, _ in res)
    rate = sum(n / dt for n, dt in res)
    return {"value": round(rate, 2), "unit": "images/s", "cores": ncores, "images": total, "wall_s": round(wall, 1)}


# ----------------------------------------------------------------------------------------------------------------
# GPU side
# ----------------------------------------------------------------------------------------------------------------
def make_inputs(B: int, S: int, dev, seed: int, layout: str):
    """Synthetic inputs of SURVEY 8d: uniform [0,1) float32 images; attention rows = softmax of a
    640-wide random-normal row, image tokens at starts = 35 + (b mod 8)."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    img = torch.rand((B, S, S, 3) if layout == "hwc" else (B, 3, S, S), device=dev, generator=g)
    rows = torch.empty((T_STEPS, B, HEADS, KV), device=dev)
    for t in range(T_STEPS):                       # chunked: keeps the temporary small
        rows[t] = torch.softmax(torch.randn((B, HEADS, KV), device=dev, generator=g), dim=-1)
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    return img, rows, starts


class Step:
    """The hot-path step on static buffers, with HIP events around the dominant kernel."""

    def __init__(self, B, S, dev, seed, mode="cv2", layout="hwc"):
        import torch
        from attwarp_amd import attention_extraction as ae, checkpoint_utils as cu, pipeline
        self.torch, self.ae, self.cu, self.pipeline = torch, ae, cu, pipeline
        self.B, self.S, self.mode, self.layout = B, S, mode, layout
        # A batch whose images + output + attention rows fit the 256 MB Infinity Cache would be served from it when
        # every step re-used the same buffers (B=64 at 336x336: 279 MB).  Small workloads therefore ROTATE over nrot
        # independent batches (>= 2 GiB in total), so that every step streams its data from HBM like a fresh batch
        # would; the 1024x1024 B=256 batch is 6.8 GB on its own (nrot = 1).
        batch_bytes = 2 * B * S * S * 3 * 4 + T_STEPS * B * HEADS * KV * 4
        self.nrot = max(1, min(8, -(-(2 << 30) // batch_bytes)))
        self.sets = []
        for i in range(self.nrot):
            img, rows, starts = make_inputs(B, S, dev, seed + 1000 * i, layout)
            self.sets.append((img, rows, torch.empty_like(img)))
        self.starts = starts
        self.img, self.rows, self.out = self.sets[0]
        self.starts_tiled = self.starts.repeat(T_STEPS)
        self.events = []
        self.k = 0

    def __call__(self, record: bool = False):
        # same three launches as attwarp_amd.pipeline.warp_from_attention_stack, with HIP events between them
        # (torch's current stream == the stream the kernels are launched on)
        torch = self.torch
        self.img, self.rows, self.out = self.sets[self.k % self.nrot]
        self.k += 1
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if record:
            ev[0].record()
        steps = self.pipeline.attention_step_maps(self.rows, self.starts, NTOK, self.starts_tiled)
        if record:
            ev[1].record()
        mx, my = self.pipeline.axis_maps_from_attention_steps(steps, (self.S, self.S))
        if record:
            ev[2].record()
        self.cu.remap_separable(self.img, mx, my, mode=self.mode, channels_last=(self.layout == "hwc"), out=self.out)
        if record:
            ev[3].record()
            self.events.append(ev)
        return self.out

    def set_layout(self, layout):
        """Re-lay every rotating batch out as HWC / CHW (fresh output buffers)."""
        torch = self.torch
        if layout != self.layout:
            perm = (0, 3, 1, 2) if layout == "chw" else (0, 2, 3, 1)
            self.sets = [(img.permute(*perm).contiguous(), rows, None) for (img, rows, _) in self.sets]
            self.sets = [(img, rows, torch.empty_like(img)) for (img, rows, _) in self.sets]
            self.layout = layout
            self.img, self.rows, self.out = self.sets[0]

    def stage_ms(self):
        """Mean duration of the three kernels of a step: (attention reduce, maps, resample)."""
        return [float(sum(e[i].elapsed_time(e[i + 1]) for e in self.events) / len(self.events)) for i in range(3)]

    def remap_ms(self):
        return [e[2].elapsed_time(e[3]) for e in self.events]


def time_steps(step, steps, warmup, D):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize; max over ranks."""
    import torch
    step.events = []
    for _ in range(warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(record=True)
    torch.cuda.synchronize()
    D.barrier()
    wall = time.perf_counter() - t0
    return D.max_over_ranks(wall), wall


def roofline_of(step, traffic=None):
    rm = step.remap_ms()
    mean = sum(rm) / len(rm)
    alg = 2.0 * step.S * step.S * 3 * 4 * step.B
    ach = alg / (mean * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "remap_rows_kernel", "mode": step.mode, "layout": step.layout.upper(),
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "algorithmic_bytes_per_launch": alg, "kernel_ms_mean": round(mean, 4),
            "kernel_ms_min": round(min(rm), 4), "launches_timed": len(rm)}


def calibration_of(step, n: int = 20):
    """torch.add(a, 1, out=b) over the step's own image and output buffers, timed with HIP events in this process right
    after the measurement: what a plain streaming kernel reaches on THIS box for the same bytes (boxes and allocations
    differ by several per cent) -- read `roofline.frac` against `calibration.frac`."""
    import torch
    ts = []
    for i in range(n + 3):
        img, _, out = step.sets[i % step.nrot]
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); torch.add(img, 1.0, out=out); e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1))
    mean = sum(ts) / len(ts)
    alg = 2.0 * step.S * step.S * 3 * 4 * step.B
    return {"kernel": "torch.add(images, 1, out=warped): same bytes, same buffers, same process",
            "torch_add_ms": round(mean, 4), "torch_add_ms_min": round(min(ts), 4),
            "achieved": round(alg / (mean * 1e-3) / 1e9, 1), "frac": round(alg / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def load_pmc_traffic(workload: str, mode: str):
    """HBM bytes per launch of the remap kernel from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic.json, collected with this same command).  None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        return d.get(f"{workload}_{mode}", d.get(workload, {})).get("remap_rows_kernel_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1024")
    ap.add_argument("--mode", choices=["cv2", "exact"], default="cv2", help="resample arithmetic of the main line")
    ap.add_argument("--layout", choices=["hwc", "chw"], default="hwc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements (exact / CHW / 336)")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="force the GPU index (smoke-testing the N>1 path on a one-GPU box with --dist-backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: rendezvous, weight broadcast, counters and the JSON line only (CPU tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # parent: no torch import, no GPU call

    B, S, cfg_idx = WORKLOADS[args.workload]
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank_env = int(os.environ.get("RANK", "0"))
    if world_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}; start it as "
                         f"`python bench.py --gpus {args.gpus}` or under torch.distributed.run with --nproc-per-node {args.gpus}")

    # all-core CPU baseline: forked workers, so it runs BEFORE anything initialises the GPU in this process
    cpu_all = None
    want_cpu = rank_env == 0 and world_env == 1 and not args.no_cpu_baseline and not args.dry_run
    if want_cpu:
        cpu_all = cpu_baseline_all_cores(S, args.mode)

    import numpy as np
    import torch
    from attwarp_amd import dist as D, _lib
    rank, world, local = D.init(args.dist_backend if not args.dry_run else (args.dist_backend or "gloo"), args.device,
                                use_gpu=not args.dry_run)
    if args.device is not None:
        local = args.device
    _lib.load()

    if args.dry_run:
        from attwarp_amd.model import MarginalNet
        net = MarginalNet(16, 24, 8)
        nbytes = D.broadcast_module_weights(net, src=0) if world > 1 else 0
        D.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))
        wall_local = time.perf_counter() - t0
        wall = D.max_over_ranks(wall_local)
        per = D.all_gather_counters({"images": float(B * args.steps), "wall_s": wall_local})
        if rank == 0:
            print(json.dumps({"metric": "warped images/sec", "value": None, "unit": "images/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": "weak",
                              "weights_broadcast": {"bytes": nbytes}, "per_rank_images": per["images"],
                              "max_wall_s": wall, "config": {"workload": args.workload, "batch_per_gpu": B}}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
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

    step = Step(B, S, dev, seed=1234 + rank, mode=args.mode, layout=args.layout)
    # device pre-conditioning, untimed and outside the contract's W warm-up steps: the first ~20 launches after start-up
    # run 1-2 % slower (clock ramp; rocprofv3 shows their maxima 15 % above the mean), which would only penalise whichever
    # measurement comes first.  The timed region below is still W warm-up steps + exactly K timed steps.
    for _ in range(PREWARM_STEPS):
        step()
    torch.cuda.synchronize()
    wall, wall_local = time_steps(step, args.steps, args.warmup, D)
    per_rank = D.all_gather_counters({"images_per_s": B * args.steps / wall_local})
    ms_per_step = wall / args.steps * 1e3
    value = world * B * args.steps / wall
    roof = roofline_of(step, load_pmc_traffic(args.workload, args.mode))
    if roof["traffic"] is not None:                 # a committed constant, not a counter read in this run
        roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on an earlier lease)"
    roof["calibration"] = calibration_of(step)

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
        "prewarm_steps": PREWARM_STEPS,
        "config": {"workload": f"batch-{B} {S}x{S}x3 float32 {args.layout.upper()} images per GPU + attention rows "
                               f"[T={T_STEPS},B,{HEADS},{KV}] float32 -> reduce -> 24x24 -> marginals -> CDF -> "
                               f"inverse maps -> bilinear warp, mode={args.mode} "
                               f"({'cv2.remap arithmetic: 1/32-px coordinates, 4 table weights' if args.mode == 'cv2' else 'unquantised bilinear = grid_sample'}) "
                               f"(BASELINE configs[{cfg_idx}])",
                   "mode": args.mode, "batch_per_gpu": B, "image_size": S, "layout": args.layout.upper(),
                   "global_batch": world * B, "rotating_batches": step.nrot,
                   "sharding": "contiguous image blocks per rank, no data-path collective"},
        "roofline": roof,
    }
    st_ms = step.stage_ms()
    result["stages_ms"] = {"attn_reduce_step_kernel": round(st_ms[0], 4), "axis_maps_from_steps_kernel": round(st_ms[1], 4),
                           "remap_rows_kernel": round(st_ms[2], 4)}
    if world > 1:
        result["per_rank_images_per_s"] = [round(v, 1) for v in per_rank["images_per_s"]]
    if bcast:
        result["weights_broadcast"] = bcast

    if want_cpu:
        n_max = min(B, 16 if S >= 1024 else 64)
        img = step.img[:n_max].cpu().numpy()
        if args.layout == "chw":
            img = np.ascontiguousarray(img.transpose(0, 2, 3, 1))
        n, dt, out0 = _cpu_loop(img, step.rows[:, :n_max].cpu().numpy(), step.starts[:n_max].cpu().numpy(), S, args.mode, 10.0)
        g0 = step.out[0].cpu().numpy()
        if args.layout == "chw":
            g0 = g0.transpose(1, 2, 0)
        result["cpu_baseline"] = {
            "value": round(n / dt, 3), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{n} image passes over the first {n_max} of the {B} {S}x{S} images through oracle/warp_ref.c "
                      f"(attention reduce .. remap, mode={args.mode}; right-inverse table from oracle/), 1 thread, {dt:.1f} s",
            "max_abs_diff_vs_gpu_image0": float(np.abs(g0 - out0).max()),    # the sample doubles as a parity check
            "all_cores": dict(cpu_all, kind="port",
                              sample=f"one process per host core ({cpu_all['cores']}), each cycling over 2 private synthetic "
                                     f"{S}x{S} images for ~10 s; sum of per-process rates"),
        }

    if world == 1 and not args.no_also:
        other = "exact" if args.mode == "cv2" else "cv2"
        step.mode = other                                     # same buffers, the other arithmetic
        w2, _ = time_steps(step, args.steps, args.warmup, D)
        result[f"also_{other}"] = {"workload": f"same batch, mode={other}", "value": round(B * args.steps / w2, 1),
                                   "unit": "images/s", "ms_per_step": round(w2 / args.steps * 1e3, 4),
                                   "roofline": roofline_of(step, load_pmc_traffic(args.workload, other))}
        step.mode = args.mode
        # the other layout (CHW is what warp_from_cdf_torch receives, MN/checkpoint_utils.py:152)
        lay2 = "chw" if args.layout == "hwc" else "hwc"
        step.set_layout(lay2)
        w3, _ = time_steps(step, args.steps, args.warmup, D)
        result[f"also_{lay2}"] = {"workload": f"same batch as [B,3,S,S] planar float32, mode={args.mode}" if lay2 == "chw"
                                  else f"same batch as [B,S,S,3], mode={args.mode}",
                                  "value": round(B * args.steps / w3, 1), "unit": "images/s",
                                  "ms_per_step": round(w3 / args.steps * 1e3, 4), "roofline": roofline_of(step)}
    del step
    torch.cuda.empty_cache()

    if world == 1 and args.workload == "1024" and not args.no_also:
        B2, S2, _ = WORKLOADS["336"]
        n2 = max(args.steps, 50)
        step2 = Step(B2, S2, dev, seed=99, mode=args.mode, layout=args.layout)
        w4, _ = time_steps(step2, n2, args.warmup, D)
        st2 = step2.stage_ms()
        result["also"] = {"workload": f"batch-{B2} {S2}x{S2} per GPU (BASELINE configs[1]), mode={args.mode}, eager launches, rotating over "
                                      f"{step2.nrot} independent batches (a single 279 MB batch would sit in the 256 MB Infinity Cache)",
                          "rotating_batches": step2.nrot,
                          "value": round(B2 * n2 / w4, 1), "unit": "images/s", "ms_per_step": round(w4 / n2 * 1e3, 4),
                          "stages_ms": [round(v, 4) for v in st2], "roofline": roofline_of(step2)}
        # the same workload with the resample of batch k overlapped with reduce + maps of batch k+1 (one HIP graph with
        # two branches, attwarp_amd.pipeline.OverlappedWarp): exactly n2 of each kernel inside the timed region
        from attwarp_amd import pipeline as _pl
        ow = _pl.OverlappedWarp(step2.img, step2.rows, step2.starts, channels_last=(args.layout == "hwc"), mode=args.mode)
        for _ in range(args.warmup):
            ow.prime(); ow.prime2(); ow.run(8); ow.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ow.prime()                               # reduce + maps of batch 0
        ow.prime2()                              # reduce of batch 1
        ow.run(n2 - 2)                           # R(k) || M(k+1) || A(k+2): graphs of 8 steps + single steps
        ow._maps(ow.cur, 1 - ow.cur); ow.flush(); ow.cur ^= 1; ow.flush()     # tail: R(n-2), M(n-1), R(n-1)
        torch.cuda.synchronize()
        w5 = time.perf_counter() - t0
        same = bool(torch.equal(ow.out, step2.out))
        result["also_overlapped"] = {"workload": f"batch-{B2} {S2}x{S2}, resample(k) || maps(k+1) || reduce(k+2) as three branches of "
                                                 f"one HIP graph (pipeline.OverlappedWarp), exactly {n2} of each kernel timed; ONE batch "
                                                 f"re-used by every step (static graph buffers): its 279 MB largely stay in the Infinity Cache",
                                     "value": round(B2 * n2 / w5, 1),
                                     "unit": "images/s", "ms_per_step": round(w5 / n2 * 1e3, 4),
                                     "bit_identical_to_serial": same}
        del step2, ow

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
