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

The timed step is the steady state of a batch STREAM: at 1024x1024 two launches (attention reduce of batch k+2 + map
construction of batch k+1 as one, then the resample of batch k; HIP events around the resample), at 336x336 one launch
for all three, replayed as HIP graphs.  Every step runs exactly one reduce, one map construction and one resample,
bit-identical to the serial launches; the plain three-launch step on one batch is attached as "also_eager".

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
PREWARM_STEPS = 0              # extra untimed device pre-conditioning in front of the W warm-up steps: NONE by default (--prewarm N adds it)


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
    """Start n ranks, watch ALL of them: the first rank that exits non-zero (or the time limit) ends the job at once --
    the others are killed (the exact PIDs this parent started) instead of waiting in a rendezvous or a barrier for a
    peer that will never arrive -- and the parent returns 1."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, ATTWARP_BENCH_CHILD="1")
        out = out0 if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))

    def kill_all():
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    deadline = time.monotonic() + float(os.environ.get("ATTWARP_BENCH_TIMEOUT_S", "3000"))
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            print(f"[bench] rank(s) failed (rank, exit code): {bad}; terminating the others", file=sys.stderr)
            kill_all()
            return 1
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            print("[bench] ranks did not finish in time; terminating them", file=sys.stderr)
            kill_all()
            return 1
        time.sleep(0.05)
    out0.seek(0)
    lines = [l for l in out0.read().decode().splitlines() if l.startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def pin_to_local_cores(local_rank: int, world: int):
    """Pin this rank to the host cores next to ITS GPU before torch is imported (launch threads, the RCCL proxy and the
    caching allocator then stay on that socket).  The GPU's NUMA-local cpulist is read from sysfs through the KFD
    topology (no HIP call: GPU nodes in KFD order = HIP device order without HIP_VISIBLE_DEVICES); ranks that share a
    cpulist split it evenly.  Falls back to an even split of the allowed cores.  Returns a short description."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    cpus, how = None, "even split of the allowed cores"
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = dict(l.split(None, 1) for l in open(f"{base}/{node}/properties").read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props["drm_render_minor"]))
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis:
            gpus = [gpus[int(v)] for v in vis.split(",") if v.strip().isdigit() and int(v) < len(gpus)]
        minor = gpus[local_rank]
        txt = open(f"/sys/class/drm/renderD{minor}/device/local_cpulist").read().strip()
        local = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            local.update(range(int(a), int(b or a) + 1))
        local = sorted(local & set(allowed))
        if local:
            sharers = [g for g in gpus[:world] if open(f"/sys/class/drm/renderD{g}/device/local_cpulist").read().strip() == txt]
            idx, cnt = sharers.index(minor), len(sharers)
            per = max(1, len(local) // cnt)
            cpus = local[idx * per:(idx + 1) * per] or local
            how = f"NUMA-local cores of renderD{minor} ({txt}), share {idx + 1}/{cnt}"
    except Exception:
        cpus = None
    if not cpus:
        per = max(1, len(allowed) // max(world, 1))
        cpus = allowed[(local_rank % max(world, 1)) * per:(local_rank % max(world, 1) + 1) * per] or allowed
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return f"{len(cpus)} cores [{cpus[0]}..{cpus[-1]}]: {how}"


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


class PipelinedStep:
    """The 1024-class step of a batch STREAM, two launches with HIP events around the resample: the attention reduce of
    the batch after next + the map construction of the next batch as ONE launch (attwarp_attn_reduce_and_maps), then the
    resample of the current batch -- every step still runs exactly one reduce, one map construction and one resample;
    the latency-bound map kernel and one launch boundary hide behind the reduce.  Same buffers as `Step` (one static
    batch at this size: every batch of the stream is the same data), bit-identical output."""

    def __init__(self, step):
        import torch
        from attwarp_amd import pipeline
        self.torch, self.step = torch, step
        self.ow = pipeline.OverlappedWarp([x[0] for x in step.sets], [x[1] for x in step.sets], step.starts,
                                          channels_last=(step.layout == "hwc"), mode=step.mode, pattern="am")
        self.ow.reset(); self.ow.prime(); self.ow.prime2()
        self.events = []
        self.B, self.S, self.mode, self.layout = step.B, step.S, step.mode, step.layout

    def __call__(self, record: bool = False):
        torch, ow = self.torch, self.ow
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
        rows = ow.rows[(ow.k + 2) % ow.n]
        T, B, heads, kv = rows.shape
        if record:
            ev[0].record()
        ow._am_launch(ow.cur, ow.k)
        if record:
            ev[1].record()
        ow._resample(ow.cur, ow.k % ow.n)
        if record:
            ev[2].record()
            self.events.append(ev)
        ow.cur ^= 1
        ow.k += 1
        return ow.out

    def remap_ms(self):
        return [e[1].elapsed_time(e[2]) for e in self.events]

    def stage_ms(self):
        n = len(self.events)
        return [sum(e[0].elapsed_time(e[1]) for e in self.events) / n, sum(e[1].elapsed_time(e[2]) for e in self.events) / n]


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


def pmc_traffic_source(workload: str, mode: str) -> str:
    """Where `roofline.traffic` comes from: NOT counters of this run (PMC passes need the profiler) but the committed summary
    of separate rocprofv3 --pmc passes over this same command; the lease they were collected on is recorded with them."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(f"{workload}_{mode}", {})
        return ("profiles/pmc_traffic.json: a COMMITTED constant, not a counter read in this run -- separate rocprofv3 --pmc "
                f"FETCH_SIZE / WRITE_SIZE passes of this command (lease: {e.get('lease', 'unrecorded')}; {e.get('same_lease_as', '')})")
    except Exception:
        return "profiles/pmc_traffic.json (committed constant)"


def load_pmc_traffic(workload: str, mode: str):
    """HBM bytes per launch of the remap kernel from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic.json, collected with this same command).  None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        e = d.get(f"{workload}_{mode}", d.get(workload, {}))
        return e.get("remap_rows_kernel_bytes_per_launch", e.get("bytes_per_launch"))
    except Exception:
        return None


def peaked_rows(step, torch):
    """Attention rows whose image tokens carry one 3x3 hot spot x100 per image, normalised per row."""
    T, B, Hh, kv = step.rows.shape
    rows = torch.full((T, B, Hh, kv), 1.0, device=step.rows.device)
    g = torch.Generator(device=step.rows.device).manual_seed(7)
    cy = torch.randint(1, 23, (B,), device=step.rows.device, generator=g)
    cx = torch.randint(1, 23, (B,), device=step.rows.device, generator=g)
    for b in range(B):
        st = int(step.starts[b])
        for dy in (-1, 0, 1):
            i0 = st + (int(cy[b]) + dy) * 24 + int(cx[b]) - 1
            rows[:, b, :, i0:i0 + 3] = 100.0
    return rows / rows.sum(-1, keepdim=True)


def overlapped_run(ow, K: int):
    """Exactly K of each kernel: reduce + maps of batch 0 and reduce of batch 1 serially, K-2 overlapped steps
    R(k) || M(k+1) || A(k+2) as HIP-graph replays (graphs of 8 steps + single steps), then the two tail resamples."""
    ow.reset()
    ow.prime()
    ow.prime2()
    ow.run(K - 2)
    ow.tail()


def time_overlapped(ow, K: int, W: int, D, prewarm_s: float = 0.0):
    """Graph path: one untimed pass of the same K steps (captures every graph the timed pass replays) + W further
    untimed steps, then exactly K steps between barrier + synchronize; no host call per kernel, no event records."""
    import torch
    overlapped_run(ow, K)
    if prewarm_s > 0:
        # device pre-conditioning, untimed and outside the contract's W warm-up steps (as PREWARM_STEPS of the 1024 workload):
        # the first second of launches of a process runs ~7 % slower (clock ramp; 0.246 against 0.217-0.228 ms per step at
        # B=256 336x336), which would only penalise a measurement that starts right after start-up
        t_end = time.perf_counter() + prewarm_s
        n_pre = 0
        ow.reset(); ow.prime(); ow.prime2()
        while time.perf_counter() < t_end:
            ow.run(64)
            n_pre += 64
            torch.cuda.synchronize()
        ow.tail()
        time_overlapped.prewarm_steps = n_pre
    if W > 0:
        ow.reset(); ow.prime(); ow.prime2(); ow.run(W); ow.tail()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    overlapped_run(ow, K)
    torch.cuda.synchronize()
    D.barrier()
    wall = time.perf_counter() - t0
    return D.max_over_ranks(wall), wall


def rank_fields(D, world: int, units_per_rank: float, wall_local: float, wall: float, ok: bool = True) -> dict:
    """What a secondary leg reports when it ran on EVERY rank of a process group (the main line's discipline: barrier +
    synchronize on both sides of exactly K steps, `wall` = max over ranks): the whole-job rate, every rank's own rate, the
    job's rate against N x the mean rank (1.0 = the ranks lose nothing to each other beyond their own spread), and the AND over
    ranks of the leg's own bit-identity check.  All ranks must call it (two collectives)."""
    per = D.all_gather_counters({"rate": units_per_rank / wall_local, "ok": 1.0 if ok else 0.0})
    rates = per["rate"]
    value = world * units_per_rank / wall
    return {"value": round(value, 1), "n_gpus": world, "per_rank_images_per_s": [round(v, 1) for v in rates],
            "scaling_efficiency_vs_rank_mean": round(value / (world * sum(rates) / len(rates)), 4),
            "bit_identical_to_serial": all(v == 1.0 for v in per["ok"])}


def agree(D, ok: bool) -> bool:
    """True when EVERY rank says ok -- a collective OUTSIDE any try block: construction and warm-up of a leg can fail on one
    rank only (out of memory, an ineligible shape); the ranks first agree that all of them are ready and only then enter the
    timed collectives together (or skip the leg together)."""
    return D.max_over_ranks(0.0 if ok else 1.0) == 0.0


def step_bytes(B: int, S: int, attn_esize: int = 4) -> float:
    """Algorithmic bytes of one step: the resample's 2*S*S*3*4 per image + the attention rows the reduce
    reads, T*heads*576*esize per image."""
    return float(B) * (2.0 * S * S * 3 * 4 + T_STEPS * HEADS * NTOK * attn_esize)


PREWARM_SMALL_S = 0.0          # seconds of extra untimed graph replays before the warm-up steps of the 336 workloads (--prewarm, tenths of a second)


def small_workload(B, S, dev, seed, mode, layout, K, W, D, torch, pipeline, attn_dtype=None, prewarm_s=0.0):
    """configs[1] / configs[3]'s per-rank batch: the overlapped graph path over a ring of independent batches (>= 2 GiB,
    every step streams from HBM) as the headline, the eager three-launch step with HIP events beside it.
    attn_dtype (torch.float16 / bfloat16): the attention rows in the model dtype LLaVA emits instead of float32."""
    st = Step(B, S, dev, seed=seed, mode=mode, layout=layout)
    esize = 4
    if attn_dtype is not None:
        st.sets = [(img, rows.to(attn_dtype), out) for (img, rows, out) in st.sets]
        st.img, st.rows, st.out = st.sets[0]
        esize = 2
    for _ in range(5):
        st()
    ow = pipeline.OverlappedWarp([x[0] for x in st.sets], [x[1] for x in st.sets], st.starts,
                                 channels_last=(layout == "hwc"), mode=mode)
    wall, wall_local = time_overlapped(ow, K, W, D, prewarm_s)
    # serial reference of every ring slot, AFTER the timed region: the overlapped outputs must equal it bit for bit
    same = True
    for r in range(min(ow.n, K)):                  # the slots the K timed steps wrote
        ref = pipeline.warp_from_attention_stack(st.sets[r][0], st.sets[r][1], st.starts, channels_last=(layout == "hwc"),
                                                 mode=mode)
        same = same and bool(torch.equal(ow.outs[r], ref))
        del ref
    # small batches: one launch per TWO batches of the stream (pipeline.PairedStepWarp: R(2p), R(2p+1) | M(2p+2), M(2p+3) |
    # A(2p+4), A(2p+5)) -- the launch's ramp and tail are paid once per two batches.  Measured beside the one-launch-per-batch
    # form on the same ring; the faster one is the line, the other is attached (B=256: equal within noise, not run).
    paired = None
    if B * S * S <= 64 * 336 * 336 and ow.n % 2 == 0 and K % 2 == 0 and K >= 8 and layout in ("hwc", "chw"):
        # construction and warm-up can fail on ONE rank only (out of memory, an ineligible shape): the ranks first agree --
        # a collective OUTSIDE any try -- that all of them are ready, and only then enter the timed collectives together
        pw, err = None, ""
        try:
            pw = pipeline.PairedStepWarp([x[0] for x in st.sets], [x[1] for x in st.sets], st.starts,
                                         channels_last=(layout == "hwc"), mode=mode)
            def run_pw():
                pw.reset(); pw.prime(); pw.run(K - 4); pw.tail()      # exactly K of each piece of work
            run_pw(); run_pw()
            torch.cuda.synchronize()
        except Exception as e:                     # (an ineligible shape keeps the one-launch-per-batch line)
            pw, err = None, str(e)[:200]
        if D.max_over_ranks(0.0 if pw is not None else 1.0) > 0.0:
            paired = {"unavailable": err or "another rank could not build the paired step"}
        else:
            D.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_pw()
            torch.cuda.synchronize(); D.barrier()
            wall_p_local = time.perf_counter() - t0
            wall_p = D.max_over_ranks(wall_p_local)
            same_p = True
            for r in range(min(pw.n, K)):
                ref = pipeline.warp_from_attention_stack(st.sets[r][0], st.sets[r][1], st.starts, channels_last=(layout == "hwc"),
                                                         mode=mode)
                same_p = same_p and bool(torch.equal(pw.outs[r], ref))
                del ref
            paired = {"ms_per_step": round(wall_p / K * 1e3, 4), "images_per_s": round(B * K / wall_p, 1),
                      "bit_identical_to_serial": same_p, "wall": wall_p, "wall_local": wall_p_local}
        del pw
    w_e, _ = time_steps(st, K, W, D)
    sb = step_bytes(B, S, esize)
    one = {"ms_per_step": round(wall / K * 1e3, 4), "images_per_s": round(B * K / wall, 1), "bit_identical_to_serial": same}
    path_txt = ("pipeline.OverlappedWarp, pattern '" + ow.pattern + "': resample(k) + maps(k+1) + reduce(k+2) "
                + ("as block ranges of ONE launch per step (attwarp_warp_step_fused)" if ow.pattern == "fused"
                   else "as branches of one HIP graph"))
    if paired and "wall" in paired and paired["wall"] < wall and paired["bit_identical_to_serial"]:
        wall, wall_local, same = paired["wall"], paired["wall_local"], paired["bit_identical_to_serial"]
        path_txt = ("pipeline.PairedStepWarp: resample(2p), (2p+1) + maps(2p+2), (2p+3) + reduce(2p+4), (2p+5) as block ranges of ONE "
                    "launch per TWO batches (attwarp_warp_step_fused_slots)")
    if paired:
        paired.pop("wall", None); paired.pop("wall_local", None)
    ms = wall / K * 1e3
    res = {"ms_per_step": round(ms, 4), "images_per_s": round(B * K / wall, 1),
           "step_algorithmic_bytes": sb, "step_TBps": round(sb / (ms * 1e-3) / 1e12, 3),
           "step_frac_of_hbm_peak": round(sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "rotating_batches": ow.n, "bit_identical_to_serial": same,
           "path": path_txt + f", HIP-graph replay (one host call per 8 batches), ring of {ow.n} independent batches (>= 2 GiB: every "
                   f"step streams from HBM), exactly {K} of each piece of work in the timed region",
           "one_launch_per_batch": one, "one_launch_per_two_batches": paired,
           "eager": {"ms_per_step": round(w_e / K * 1e3, 4), "images_per_s": round(B * K / w_e, 1),
                     "step_TBps": round(sb / (w_e / K) / 1e12, 3),
                     "stages_ms": [round(v, 4) for v in st.stage_ms()],
                     "roofline": roofline_of(st),
                     "path": "three eager launches per step with HIP events between them (the events are where the "
                             "per-kernel durations come from)"}}
    return res, wall, wall_local, st, ow


def traffic_note(roof):
    """`traffic` of a secondary line: a committed counter pass if profiles/pmc_traffic.json holds one for exactly this
    variant, else null WITH the reason (no silent nulls)."""
    if roof.get("traffic") is None:
        roof["traffic_source"] = "null: no rocprofv3 --pmc pass committed for this variant (profiles/pmc_traffic.json)"
    else:
        roof["traffic_source"] = ("profiles/pmc_traffic.json: a COMMITTED constant (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes; the lease is recorded in that file), not a counter read in this run")
    return roof


def _event_ms(torch, fn, n, warm=3):
    """Median duration of fn() over n runs, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def copy_calibration(torch, dev, read_bytes: int, write_bytes: int, n: int = 40):
    """The ruler of a chain step: ONE launch of a plain streaming kernel (csrc/calib.hip, tuning flavour of the library) that
    reads `read_bytes` and writes `write_bytes` -- the step's algorithmic bytes at the same batch size -- rotating over
    >= 2 GiB of source and destination so that every launch streams from HBM like the step does; mean of n launches timed as
    ONE region with HIP events (back to back, as the stream's steps run).  None when the tuning library is missing."""
    from attwarp_amd import _lib
    import ctypes
    try:
        tl = _lib.load_tuning()
    except ImportError:
        return None
    rb, wb = (int(read_bytes) + 15) // 16 * 16, (int(write_bytes) + 15) // 16 * 16
    slots = max(2, min(64, -(-(2 << 30) // (rb + wb))))
    src = torch.empty(slots * rb, device=dev, dtype=torch.uint8).random_(0, 256)
    dst = torch.empty(slots * wb, device=dev, dtype=torch.uint8)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    def run(k):
        for i in range(k):
            j = i % slots
            rc = tl.attwarp_debug_stream_copy(ctypes.c_void_p(src.data_ptr() + j * rb), rb, ctypes.c_void_p(dst.data_ptr() + j * wb), wb, st)
            if rc != 0:
                raise RuntimeError(tl.attwarp_last_error().decode())
    run(8)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run(n); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    del src, dst
    return {"kernel": "debug_stream_copy_kernel (csrc/calib.hip): one launch, reads read_bytes + writes write_bytes, rotating over "
                      f"{slots} slots (>= 2 GiB)", "read_bytes": int(read_bytes), "write_bytes": int(write_bytes),
            "ms": round(ms, 4), "frac_of_hbm_peak": round((read_bytes + write_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def leg_main_batched(dev, torch, pipeline, K):
    """The chain the reference's batched driver runs per image (AGW/main_batched.py:243-287, BATCH_SIZE = 32 at :42, 500 x 500
    output at :62-63): revise_mask -> x255 uint8 -> PIL LANCZOS -> float64 marginals -> CDF -> np.interp -> uint8 cv2.remap,
    on uint8 [B,S,S,3] images + [B,24,24] float32 attention maps.  Per case: `serial` = pipeline.warp_from_masks (five
    dependent launches per batch, eager), `stream` = pipeline.MaskChainStream (the same five stages on five consecutive
    batches, graph replay), both over a ring of independent batches of >= 2 GiB (every step streams from HBM), exactly K
    steps in the timed region.  Chain-level algorithmic bytes per image: S*S mask written + S*S mask read by
    the marginals + 3*S*S image read + 3*So*So written."""
    K = max(K + (K & 1), 10)        # the stream needs K >= its pipeline depth (two batches per launch: K / 2 >= depth)
    out = {"workload": "uint8 images [B,S,S,3] + attention maps [B,24,24] -> main_batched chain -> [B,500,500,3] uint8, mode=cv2, "
                       "transform=identity (main_batched.py:280-287)", "unit": "images/s", "steps": K, "cases": []}
    for (B, S, So) in ((32, 336, 500), (64, 336, 500), (256, 336, 500), (32, 1024, 500), (64, 1024, 500), (256, 1024, 500)):
        slot = B * (3 * S * S + 3 * So * So)
        n = max(6, min(64, -(-(2 << 30) // slot)))
        n += n & 1
        g = torch.Generator(device=dev).manual_seed(77 + B + S)
        # (each ring is ONE allocation, so that two consecutive slots can also be run as one batch of 2B: pipeline.pair_slots)
        images = list(torch.randint(0, 256, (n, B, S, S, 3), device=dev, dtype=torch.uint8, generator=g))
        masks = list(torch.rand(n, B, 24, 24, device=dev, generator=g))
        bytes_img = 2 * S * S + 3 * S * S + 3 * So * So
        # serial: the drop-in itself, batch after batch over the ring
        for i in range(3):
            pipeline.warp_from_masks(images[i % n], masks[i % n], (So, So))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            o = pipeline.warp_from_masks(images[i % n], masks[i % n], (So, So))
        torch.cuda.synchronize()
        t_serial = (time.perf_counter() - t0) / K
        del o
        # stream
        mc = pipeline.MaskChainStream(images, masks, (So, So))
        def run():
            mc.reset(); mc.prime(); mc.run(K - mc.depth); mc.drain()      # exactly K of every stage
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        t_stream = (time.perf_counter() - t0) / K
        same = all(bool(torch.equal(mc.outs[i], pipeline.warp_from_masks(images[i], masks[i], (So, So)))) for i in range(min(n, K)))
        # small batches: two batches of the stream per launch (slots 2i, 2i+1 as one batch of 2B through the same kernel)
        two = None
        if B <= 64 and K % 2 == 0:
            mc2 = pipeline.MaskChainStream(pipeline.pair_slots(images), pipeline.pair_slots(masks), (So, So))
            def run2():
                mc2.reset(); mc2.prime(); mc2.run(K // 2 - mc2.depth); mc2.drain()      # exactly K batches through every stage
            run2(); run2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run2()
            torch.cuda.synchronize()
            t_two = (time.perf_counter() - t0) / K
            same2 = all(bool(torch.equal(mc2.outs[i // 2][(i % 2) * B:(i % 2 + 1) * B], mc.outs[i])) for i in range(min(n, K)))
            two = {"ms_per_batch": round(t_two * 1e3, 4), "images_per_s": round(B / t_two, 1),
                   "step_frac_of_hbm_peak": round(B * bytes_img / t_two / 1e9 / HBM_PEAK_GBS, 4), "bit_identical_to_serial": same2}
            del mc2
        calib = copy_calibration(torch, dev, B * 5 * S * S, B * 3 * So * So)
        if calib:
            calib["step_over_copy"] = round(t_stream * 1e3 / calib["ms"], 3)
            if two:
                calib["two_batches_per_launch_over_copy"] = round(two["ms_per_batch"] / calib["ms"], 3)
        case = {"B": B, "S": S, "S_out": So, "ring_batches": n, "pattern": mc.pattern, "two_batches_per_launch": two,
                "calibration": calib,
                "chain_algorithmic_bytes_per_image": bytes_img,
                "stream": {"ms_per_step": round(t_stream * 1e3, 4), "images_per_s": round(B / t_stream, 1),
                           "step_TBps": round(B * bytes_img / t_stream / 1e12, 3),
                           "step_frac_of_hbm_peak": round(B * bytes_img / t_stream / 1e9 / HBM_PEAK_GBS, 4)},
                "serial": {"ms_per_step": round(t_serial * 1e3, 4), "images_per_s": round(B / t_serial, 1),
                           "step_frac_of_hbm_peak": round(B * bytes_img / t_serial / 1e9 / HBM_PEAK_GBS, 4)},
                "bit_identical_to_serial": same}
        out["cases"].append(case)
        del mc, images, masks
        torch.cuda.empty_cache()
    ref = [c for c in out["cases"] if (c["B"], c["S"]) == (32, 336)][0]
    out["value"] = ref["stream"]["images_per_s"]
    out["value_is"] = ("the stream step at the reference's own scale: B=32, 336 -> 500, one batch per launch "
                       "(`two_batches_per_launch` beside it)")
    out["step_over_copy"] = (ref.get("calibration") or {}).get("step_over_copy")
    out["step_over_copy_is"] = ("stream ms per step / the ms of ONE plain streaming launch that reads 5*S*S and writes 3*S_out*S_out bytes "
                                "per image at the same B (`calibration` of each case): what the chain costs over a copy of its bytes")
    return out


TEXTVQA_LIKE_WH = [(1024, 768), (683, 1024), (1024, 1024), (500, 375), (333, 500), (640, 427)]      # W x H as PIL reports them


def leg_main_batched_ragged(dev, torch, pipeline, K, D=None, rank=0, world=1, grouped=False):
    """The same chain on what the reference's driver actually holds (AGW/main_batched.py:243-287): batches of DIFFERENTLY
    sized images (`b_images[j]` at native size; a TextVQA-like mix of 1024 x 768, 683 x 1024, 1024 x 1024, 500 x 375,
    333 x 500, 640 x 427), every mask up-sampled to its image's own size, dense [B,500,500,3] output.  Per case:
      `stream`     pipeline.RaggedMaskChainStream over a ring of prebuilt batches: ONE launch per batch (R(k) | F(k+1) |
                   P(k+2) | L(k+3) | V(k+4)), eager and replayed as HIP graphs -- timed between barrier + synchronize, max over
                   ranks; in a process group EVERY rank runs it on its own batches (per-rank rates reported);
      `calibration`  one plain streaming launch of the same bytes (copy_calibration) and `step_over_copy`;
    and, on one GPU without a process group, beside it:
      `per_image`  the loop a driver had to write before: pipeline.warp_from_masks on one image at a time (5 launches each);
      `serial`     pipeline.warp_from_masks_ragged: table build + upload + five ragged launches per batch;
      `equal_size_stream`  pipeline.MaskChainStream on B images of S x S with S*S = the mix's mean pixel count (the
                   one-launch step for equally sized images): the bar is ragged <= 1.3 x that.
    Rings of >= 2 GiB of independent batches, exactly K batches through every stage in each timed region; chain-level
    algorithmic bytes as `also_main_batched`."""
    from attwarp_amd import dist as D0
    D = D or D0
    K = max(K + (K & 1), 10)
    So = 500
    full = world == 1 and not grouped
    out = {"workload": "uint8 images of different sizes [H_i,W_i,3] + attention maps [B,24,24] -> main_batched chain -> "
                       "[B,500,500,3] uint8, mode=cv2, transform=identity", "unit": "images/s", "steps": K,
           "sizes_WxH": TEXTVQA_LIKE_WH, "cases": []}
    for B in ((32, 64, 256) if full else (32, 256)):   # (64: two of the driver's batches of 32 as ONE ragged batch -- any sizes mix)
        sizes = [TEXTVQA_LIKE_WH[b % len(TEXTVQA_LIKE_WH)] for b in range(B)]
        px = sum(w * h for (w, h) in sizes)
        slot = 3 * px + 3 * B * So * So
        n = max(6, min(40, -(-(2 << 30) // slot)))
        bytes_batch = 5 * px + 3 * B * So * So
        st = ring = ring_imgs = masks = None
        err = ""
        try:
            g = torch.Generator(device=dev).manual_seed(99 + B + 1000 * rank)
            ring_imgs = [[torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (w, h) in sizes]
                         for _ in range(n)]
            masks = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
            ring = []
            for i in range(n):
                rb = pipeline.RaggedBatch(ring_imgs[i], (So, So)); rb.masks = masks[i]
                ring.append(rb)
            st = pipeline.RaggedMaskChainStream(out_size=(So, So))
            st.ring(ring)
            def run(unroll):
                st.k = 0
                st.prime(); st.run(K - st.DEPTH, unroll=unroll); st.drain_ring()      # exactly K batches through every stage
            for unroll in (0, n):                      # warm-up: every graph the timed passes replay is captured here
                run(unroll); run(unroll)
            torch.cuda.synchronize()
        except Exception as e:                         # noqa: BLE001 -- one rank only may fail (memory): agree below
            st, err = None, f"{type(e).__name__}: {str(e)[:200]}"
        if not agree(D, st is not None):
            out["cases"].append({"B": B, "unavailable": err or "another rank could not build this case"})
            del st, ring, ring_imgs, masks
            torch.cuda.empty_cache()
            continue
        res, walls = {}, {}
        for name, unroll in (("eager", 0), ("graphs", n)):
            D.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(unroll)
            torch.cuda.synchronize(); D.barrier()
            wl = time.perf_counter() - t0
            w = D.max_over_ranks(wl)
            walls[name] = (w, wl)
            res[name] = {"ms_per_batch": round(w / K * 1e3, 4), "images_per_s": round(B * K / w, 1),
                         "step_frac_of_hbm_peak": round(bytes_batch / (w / K) / 1e9 / HBM_PEAK_GBS, 4)}
        want0 = pipeline.warp_from_masks_ragged(ring_imgs[0], masks[0], (So, So))
        same = bool(torch.equal(ring[0].out, want0)) and all(
            bool(torch.equal(ring[i].out, pipeline.warp_from_masks_ragged(ring_imgs[i], masks[i], (So, So)))) for i in range(1, min(n, 4)))
        best = "graphs" if walls["graphs"][0] <= walls["eager"][0] else "eager"
        case = {"B": B, "pixels_per_batch": px, "ring_batches": n, "chain_algorithmic_bytes_per_batch": bytes_batch,
                "stream": res, "bit_identical_to_serial": same}
        if grouped:
            case.update(rank_fields(D, world, B * K, walls[best][1], walls[best][0], same))
            case["value_is"] = f"whole-job images/s of the `{best}` stream (weak scaling: every rank its own batches of {B})"
        calib = copy_calibration(torch, dev, 5 * px, 3 * B * So * So)
        if calib:
            calib["step_over_copy"] = round(res[best]["ms_per_batch"] / calib["ms"], 3)
        case["calibration"] = calib
        del ring, st
        if full:
            # the drop-in for one batch at a time (table build + upload + five launches)
            for i in range(2):
                pipeline.warp_from_masks_ragged(ring_imgs[i % n], masks[i % n], (So, So))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                o = pipeline.warp_from_masks_ragged(ring_imgs[i % n], masks[i % n], (So, So))
            torch.cuda.synchronize()
            t_serial = (time.perf_counter() - t0) / K
            # what a driver had to do without it: image by image (timed on a bounded sample of batches)
            kk = 2 if B > 64 else 4
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(kk):
                for b in range(B):
                    o = pipeline.warp_from_masks(ring_imgs[i % n][b][None], masks[i % n][b:b + 1], (So, So))
            torch.cuda.synchronize()
            t_per_image = (time.perf_counter() - t0) / kk
            del o
            # the equal-size stream step at the same total pixels
            S = int(round((px / B) ** 0.5 / 4)) * 4
            ne = n + (n & 1)
            images_e = list(torch.randint(0, 256, (ne, B, S, S, 3), device=dev, dtype=torch.uint8, generator=g))
            masks_e = list(torch.rand(ne, B, 24, 24, device=dev, generator=g))
            mc = pipeline.MaskChainStream(images_e, masks_e, (So, So))
            def run_e():
                mc.reset(); mc.prime(); mc.run(K - mc.depth); mc.drain()
            run_e(); run_e()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_e()
            torch.cuda.synchronize()
            t_eq = (time.perf_counter() - t0) / K
            case.update({
                "serial": {"ms_per_batch": round(t_serial * 1e3, 4), "images_per_s": round(B / t_serial, 1),
                           "includes": "table build on the host, its upload, buffer allocation, five launches"},
                "per_image": {"ms_per_batch": round(t_per_image * 1e3, 4), "images_per_s": round(B / t_per_image, 1)},
                "equal_size_stream": {"S": S, "pattern": mc.pattern, "ms_per_batch": round(t_eq * 1e3, 4),
                                      "images_per_s": round(B / t_eq, 1)},
                "ragged_over_equal_size": round(res[best]["ms_per_batch"] / (t_eq * 1e3), 3)})
            del mc, images_e, masks_e
        out["cases"].append(case)
        del ring_imgs, masks
        torch.cuda.empty_cache()
    ref = next((c for c in out["cases"] if c["B"] == 32 and "stream" in c), None)
    if ref is None:
        out["value"] = None
        return out
    if grouped:
        for k in ("value", "n_gpus", "per_rank_images_per_s", "scaling_efficiency_vs_rank_mean", "bit_identical_to_serial"):
            out[k] = ref[k]
        out["value_is"] = "whole-job rate of the ragged stream step at the reference's batch size (B=32 per rank), one launch per batch"
    else:
        out["value"] = max(ref["stream"]["eager"]["images_per_s"], ref["stream"]["graphs"]["images_per_s"])
        out["value_is"] = "the ragged stream step at the reference's batch size (B=32), one launch per batch"
    out["step_over_copy"] = (ref.get("calibration") or {}).get("step_over_copy")
    if rank == 0 and full:
        out["cpu_baseline"] = chain_cpu_baseline(dev, torch, pipeline, So)
    return out


def chain_cpu_baseline(dev, torch, pipeline, So, budget_s: float = 8.0):
    """The reference's CPU chain for the same images, timed beside the GPU path (kind "port": oracle/warp_oracle.py -- numpy --
    for revise_mask, the x255 truncation, Pillow's LANCZOS arithmetic and the float64 map construction, oracle/warp_ref.c for
    the uint8 cv2 resample), one thread, on a BOUNDED sample: the first images of the TextVQA-like mix, cycled until the budget
    is spent.  Doubles as a check: every image it processes is compared with the GPU's output for that image."""
    import numpy as np
    from oracle import c_oracle, warp_oracle as O
    sizes = [TEXTVQA_LIKE_WH[b % len(TEXTVQA_LIKE_WH)] for b in range(6)]
    g = torch.Generator(device=dev).manual_seed(4242)
    imgs = [torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (w, h) in sizes]
    att = torch.rand(len(imgs), 24, 24, device=dev, generator=g) ** 2
    gpu = pipeline.warp_from_masks_ragged(imgs, att, (So, So)).cpu().numpy()
    imgs_h = [i.cpu().numpy() for i in imgs]
    att_h = att.cpu().numpy()
    worst, n, t0 = 0, 0, time.perf_counter()
    while True:
        b = n % len(imgs_h)
        h, w = imgs_h[b].shape[:2]
        with np.errstate(all="ignore"):
            mota = O.lanczos_resize_u8(O.mask_to_u8(O.revise_mask(att_h[b], 3, 10)), w, h)
        mx, my = O.maps_from_attention(mota, So, So, "identity")
        ref = c_oracle.remap_bilinear_u8(imgs_h[b], mx, my, "cv2")
        if n < len(imgs_h):
            worst = max(worst, int(np.abs(ref.astype(np.int16) - gpu[b].astype(np.int16)).max()))
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= len(imgs_h):
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{n} image passes over {len(imgs_h)} images of the mix {sizes} (W x H) -> {So} x {So} through the oracle's chain "
                      f"(numpy revise_mask / x255 / LANCZOS / float64 maps + oracle/warp_ref.c resample), 1 thread, {dt:.1f} s",
            "max_abs_diff_vs_gpu_grey_levels": worst,
            "note": "the GPU's revised mask may differ from the oracle's by 1 ulp, which can move a x255-truncated mask cell by one "
                    "grey level (DESIGN 4): a difference of at most 1 grey level in a few pixels is expected, bit-exactness GIVEN the "
                    "revised mask is what the tests assert"}


def leg_u8(dev, torch, pipeline, K):
    """The resample the reference's drivers actually run: uint8 BGR in, uint8 out (AGW/new_method.py:268-271), the integer cv2
    kernel (`remap_rows_u8i_kernel`) ALONE -- maps from a mildly peaked 24-bin PDF, HIP events around every launch, rotating
    over >= 2 GiB of independent batches so every launch streams from HBM.  `roofline` per case on the algorithmic bytes
    (S_in*W_in + S_out*S_out) * 3 per image (uint8 variant); `unaligned_over_aligned`: 683-pixel-wide rows (2049
    bytes: the portrait TextVQA case, the kernel's unaligned form) against 684-pixel-wide ones."""
    from attwarp_amd import checkpoint_utils as cu
    out = {"workload": "uint8 [B,H,W,3] -> [B,S_out,S_out,3], mode=cv2: the integer resample alone", "unit": "images/s", "cases": []}
    ms_of = {}
    for (B, H, W, So) in ((256, 1024, 1024, 1024), (256, 1024, 1024, 500), (64, 336, 336, 500), (256, 1024, 684, 500), (256, 1024, 683, 500)):
        per = B * (H * W * 3 + So * So * 3)
        n = max(1, min(16, -(-(2 << 30) // per)))
        g = torch.Generator(device=dev).manual_seed(B + W + So)
        imgs = [torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
        outs = [torch.empty(B, So, So, 3, device=dev, dtype=torch.uint8) for _ in range(n)]
        px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.3, 1)
        py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.3, 1)
        mx, my = pipeline.axis_maps_from_pdf(px, py, (H, W), (So, So))
        for i in range(3):
            cu.remap_separable(imgs[i % n], mx, my, mode="cv2", channels_last=True, out=outs[i % n])
        torch.cuda.synchronize()
        evs = []
        for i in range(K):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); cu.remap_separable(imgs[i % n], mx, my, mode="cv2", channels_last=True, out=outs[i % n]); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in evs]
        ms = sum(ts) / len(ts)
        ms_of[(B, H, W, So)] = ms
        ach = per / (ms * 1e-3) / 1e9
        out["cases"].append({"B": B, "H": H, "W": W, "S_out": So, "rotating_batches": n, "images_per_s": round(B / (ms * 1e-3), 1),
                             "roofline": {"bound": "hbm", "kernel": "remap_rows_u8i_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": per,
                                          "kernel_ms_mean": round(ms, 4), "kernel_ms_min": round(min(ts), 4), "launches_timed": K,
                                          "traffic": None,
                                          "traffic_source": "null: this kernel loads ONE dword per lane, an access width for which FETCH_SIZE is "
                                                            "uncalibrated on gfx950 (MI355X_MICROARCH.md, HBM); the raw FETCH_SIZE / WRITE_SIZE passes "
                                                            "are in profiles/round6_u8_pmc_*.txt (1024 -> 1024: 2 x FETCH + WRITE = 1.028 x algorithmic)"}})
        del imgs, outs
        torch.cuda.empty_cache()
    out["unaligned_over_aligned"] = round(ms_of[(256, 1024, 683, 500)] / ms_of[(256, 1024, 684, 500)], 3)
    out["value"] = out["cases"][0]["images_per_s"]
    out["value_is"] = "B=256, 1024 x 1024 x 3 -> 1024 x 1024 x 3 (the headline size in the dtype the reference's drivers use)"
    out["roofline"] = out["cases"][0]["roofline"]
    return out


def leg_pool_input(dev, torch, pipeline, B, S, mode, K):
    """SURVEY 8d config 3's other input form: full-resolution attention [B,1,S,S] float32 -> F.adaptive_avg_pool2d to 24 x 24
    (MN/trainer.py:197 + sanitise :202) -> gt_marginals (MN/checkpoint_utils.py:43-51) -> right-inverse PDF -> CDF ->
    inverse maps -> warp of [B,S,S,3] float32, stage by stage with HIP events."""
    from attwarp_amd import checkpoint_utils as cu
    g = torch.Generator(device=dev).manual_seed(2)
    A = torch.rand((B, 1, S, S), device=dev, generator=g)
    img = torch.rand((B, S, S, 3), device=dev, generator=g)
    outb = torch.empty_like(img)
    st = {}
    a24 = pipeline.adaptive_avg_pool2d(A, (24, 24), sanitize=True)
    px, py = cu.gt_marginals(a24)
    mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
    st["adaptive_pool_kernel"] = _event_ms(torch, lambda: pipeline.adaptive_avg_pool2d(A, (24, 24), sanitize=True), K)
    st["gt_marginals_24x24"] = _event_ms(torch, lambda: cu.gt_marginals(a24), K)
    st["axis_maps_from_pdf_kernel"] = _event_ms(torch, lambda: pipeline.axis_maps_from_pdf(px, py, (S, S)), K)
    st["remap_rows_kernel"] = _event_ms(torch, lambda: cu.remap_separable(img, mx, my, mode=mode, channels_last=True, out=outb), K)
    def whole():
        a = pipeline.adaptive_avg_pool2d(A, (24, 24), sanitize=True)
        p, q = cu.gt_marginals(a)
        x, y = pipeline.axis_maps_from_pdf(p, q, (S, S))
        cu.remap_separable(img, x, y, mode=mode, channels_last=True, out=outb)
    ms = _event_ms(torch, whole, K)
    pool_bytes = 4.0 * S * S * B
    return {"workload": f"full-resolution attention [{B},1,{S},{S}] float32 -> 24x24 pool -> marginals -> maps -> warp of [{B},{S},{S},3] "
                        f"float32, mode={mode}", "value": round(B / (ms * 1e-3), 1), "unit": "images/s",
            "ms_per_step": round(ms, 4), "stages_ms": {k: round(v, 4) for k, v in st.items()},
            "roofline_pool": {"bound": "hbm", "kernel": "adaptive_pool_kernel", "algorithmic_bytes_per_launch": pool_bytes,
                              "achieved": round(pool_bytes / (st["adaptive_pool_kernel"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(pool_bytes / (st["adaptive_pool_kernel"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}


def leg_config5(dev, torch, pipeline, B, K):
    """BASELINE configs[4] as a device-resident data flow (pipeline.config5_chain): per-leg milliseconds, so the share of the
    hand-written path in an end-to-end step is a number.  The vision tower has the architecture of LLaVA-1.5's
    (CLIP ViT-L/14-336) with seeded RANDOM weights; TextVQA accuracy parity is unobtainable here (no weights, no llava
    package, no dataset)."""
    from attwarp_amd.model import MarginalNet
    tower = pipeline.random_clip_vision_tower(dev, torch.float16, seed=0)
    torch.manual_seed(5)
    net = MarginalNet(1024, 4096, 256).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(55)
    imgs = torch.randint(0, 256, (B, 500, 500, 3), device=dev, dtype=torch.uint8, generator=g)
    txt = torch.randn(B, 32, 4096, device=dev, generator=g)
    mask = torch.ones(B, 32, 1, device=dev)
    legs = ["clip_tensor_in", "tower_in", "marginalnet", "warp", "clip_tensor_warped", "tower_warped"]
    acc = {k: 0.0 for k in legs}
    total = 0.0
    for it in range(K + 2):
        evs = [torch.cuda.Event(enable_timing=True)]
        evs[0].record()
        def rec(name):
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        r = pipeline.config5_chain(tower, net, imgs, txt, mask, (500, 500), record=rec)
        torch.cuda.synchronize()
        if it >= 2:
            for i, k in enumerate(legs):
                acc[k] += evs[i].elapsed_time(evs[i + 1])
            total += evs[0].elapsed_time(evs[-1])
    legs_ms = {k: round(v / K, 4) for k, v in acc.items()}
    ms = total / K
    path_ms = legs_ms["clip_tensor_in"] + legs_ms["warp"] + legs_ms["clip_tensor_warped"]
    finite = bool(torch.isfinite(r["features_warped"].float()).all())
    del tower, net
    torch.cuda.empty_cache()
    return {"workload": f"configs[4] data flow, B={B}: uint8 500x500 images -> CLIP tensor -> ViT-L/14-336 tower (random weights, fp16) -> "
                        "[B,1024,24,24] -> MarginalNet(1024,4096,256) -> px,py -> warp -> CLIP tensor -> tower",
            "value": round(B / (ms * 1e-3), 1), "unit": "images/s", "ms_per_step": round(ms, 4), "legs_ms": legs_ms,
            "hand_written_path_ms": round(path_ms, 4), "hand_written_path_share": round(path_ms / ms, 4),
            "features_finite": finite,
            "accuracy_parity": "unobtainable here: LLaVA-1.5-7B weights, the llava package and TextVQA are absent (no network)"}


def leg_other_mode(step, args, D, torch, pipeline, B):
    """The main batch in the other arithmetic mode (same buffers)."""
    other = "exact" if args.mode == "cv2" else "cv2"
    step.mode = other
    w2, _ = time_steps(step, args.steps, args.warmup, D)
    out = {f"also_{other}": {"workload": f"same batch, mode={other}", "value": round(B * args.steps / w2, 1),
                             "unit": "images/s", "ms_per_step": round(w2 / args.steps * 1e3, 4),
                             "roofline": traffic_note(roofline_of(step, load_pmc_traffic(args.workload, other)))}}
    step.mode = args.mode
    return out


def leg_other_layout(step, args, D, torch, pipeline, B):
    """The main batch in the other layout (CHW is what warp_from_cdf_torch receives, MN/checkpoint_utils.py:152)."""
    lay2 = "chw" if args.layout == "hwc" else "hwc"
    step.set_layout(lay2)
    w3, _ = time_steps(step, args.steps, args.warmup, D)
    out = {f"also_{lay2}": {"workload": f"same batch as [B,3,S,S] planar float32, mode={args.mode}" if lay2 == "chw"
                            else f"same batch as [B,S,S,3], mode={args.mode}",
                            "value": round(B * args.steps / w3, 1), "unit": "images/s",
                            "ms_per_step": round(w3 / args.steps * 1e3, 4),
                            "roofline": traffic_note(roofline_of(step, load_pmc_traffic(args.workload, f"{args.mode}_{lay2}")))}}
    step.set_layout(args.layout)
    return out


def leg_fused_1024(step, args, D, torch, pipeline, B):
    """The 1024 step as ONE launch (attwarp_warp_step_fused through pipeline.OverlappedWarp): at this size it only hides the
    map construction and two launch boundaries behind the resample."""
    hwc = args.layout == "hwc"
    ow = pipeline.OverlappedWarp([x[0] for x in step.sets], [x[1] for x in step.sets], step.starts,
                                 channels_last=hwc, mode=args.mode, pattern="fused")
    w5, _ = time_overlapped(ow, args.steps, args.warmup, D)
    same = bool(torch.equal(ow.outs[0], pipeline.warp_from_attention_stack(step.sets[0][0], step.sets[0][1], step.starts,
                                                                        channels_last=hwc, mode=args.mode)))
    out = {"also_fused": {"workload": "same batch; reduce + maps + resample of a step as one launch (pattern "
                                      f"'{ow.pattern}'), HIP-graph replay, exactly {args.steps} of each kernel",
                          "value": round(B * args.steps / w5, 1), "unit": "images/s",
                          "ms_per_step": round(w5 / args.steps * 1e3, 4), "bit_identical_to_serial": same,
                          "step_TBps": round(step_bytes(B, step.S) / (w5 / args.steps) / 1e12, 3)}}
    del ow
    torch.cuda.empty_cache()
    return out


def leg_distributions(step, args, D, torch, pipeline, B):
    """SURVEY 8d "value distributions to also run": peaked attention (one 3x3 hot spot x100: strong magnification there,
    minification elsewhere) and all-zero attention (the uniform fallback, AGW/new_method.py:231-239 / clamp_min(1e-6) in
    MN/checkpoint_utils.py:36) on the same images."""
    out = {}
    for name, rows in (("peaked", peaked_rows(step, torch)), ("zero_attention", torch.zeros_like(step.rows))):
        keep = step.sets
        step.sets = [(img, rows, o) for (img, _, o) in keep]
        w4, _ = time_steps(step, args.steps, args.warmup, D)
        note = ("all-zero attention collapses the CDF (clamp_min(1e-6), MN/checkpoint_utils.py:36-40): every output pixel "
                "samples the LAST source row and column, so the kernel reads two source rows per image and is bound by its "
                "writes -- `roofline` therefore counts the bytes actually moved (the S*S*3*4 written per image), not the nominal "
                "2*S*S*3*4") if name == "zero_attention" else \
               "one 3x3 hot spot x100 per image: magnified there, minified elsewhere (two source rows per output row)"
        roof = traffic_note(roofline_of(step, load_pmc_traffic(args.workload, f"{args.mode}_{name}")))
        if name == "zero_attention":
            moved = 1.0 * step.S * step.S * 3 * 4 * step.B
            ach = moved / (roof["kernel_ms_mean"] * 1e-3) / 1e9
            roof.update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": moved,
                         "bytes_are": "writes only: S*S*3*4 per image (the two source rows read per image are < 0.2 %)"})
        out[f"also_{name}"] = {"workload": f"same images, {name.replace('_', ' ')} rows, mode={args.mode}", "note": note,
                               "value": round(B * args.steps / w4, 1), "unit": "images/s",
                               "ms_per_step": round(w4 / args.steps * 1e3, 4), "roofline": roof}
        step.sets = keep
        del rows
    return out


def leg_small(wl, dev, args, K, D, torch, pipeline, attn_dtype=None, rank=0, world=1, grouped=False):
    """BASELINE configs[1] / configs[3]'s per-rank batch as a secondary line of the 1024 run (graph-replayed stream step).
    In a process group EVERY rank runs it on its own batches (weak scaling: `value` is the whole job's rate)."""
    B2, S2, cfg2 = WORKLOADS[wl]
    res, wall, wall_local, st, ow = small_workload(B2, S2, dev, 99 + rank, args.mode, args.layout, K, args.warmup, D, torch,
                                                   pipeline, attn_dtype=attn_dtype)
    what = (f"batch-{B2} {S2}x{S2} per GPU (BASELINE configs[{cfg2}]), mode={args.mode}" if attn_dtype is None else
            f"batch-{B2} {S2}x{S2} per GPU, attention rows float16 (images float32), mode={args.mode}")
    out = dict({"workload": what, "value": res["images_per_s"], "unit": "images/s", "steps": K}, **res)
    if grouped:
        out.update(rank_fields(D, world, B2 * K, wall_local, wall, res["bit_identical_to_serial"]))
        out["images_per_s_per_gpu"] = out.pop("images_per_s")
        out["global_batch"] = world * B2
    del st, ow
    torch.cuda.empty_cache()
    return out


SCALING_CURVE_NOTE = ("one point per run: `value` is the whole-job rate at n_gpus = N (and `also_336x256` -- BASELINE configs[3], 256 images "
                      "per rank -- and `also_main_batched_ragged` carry theirs); the 1/2/4/8 curve and its efficiency are computed by the "
                      "driver from the per-N lines, never reported here")


def dist_leg_plan(workload: str, want, grouped: bool):
    """(key, images per rank per step) of the secondary legs that run on EVERY rank of a process group behind the main 1024
    line: BASELINE configs[3] (batch-2048 at 336 x 336 over 8 ranks = 256 per rank, the only config BASELINE names for the
    scaling curve) and the reference driver's own uint8 chain on differently sized images."""
    if workload != "1024" or not grouped:
        return []
    plan = []
    if "336" in want:
        plan.append(("also_336x256", 256))
    if "main_batched_ragged" in want:
        plan.append(("also_main_batched_ragged", 32))
    return plan


# secondary measurements attached to the default (1024) line on one GPU; `--legs a,b` selects, `--list-legs` prints
LEGS = {
    "exact": "the main batch in the other arithmetic mode (also_exact / also_cv2)",
    "chw": "the main batch in the other layout (also_chw / also_hwc)",
    "fused": "the 1024 step as ONE launch (also_fused)",
    "distributions": "SURVEY 8d value distributions: peaked and all-zero attention (also_peaked, also_zero_attention)",
    "336": "BASELINE configs[1] and configs[3]'s per-rank batch as graph-replayed one-launch steps (also, also_336x256)",
    "fp16_attention": "configs[3]'s per-rank batch with float16 attention rows (also_336x256_fp16_attention)",
    "main_batched": "the reference's own uint8 chain (main_batched.py:243-287) as a stream step vs its serial launches (also_main_batched)",
    "main_batched_ragged": "the same chain on batches of DIFFERENTLY sized images, as that driver holds them: one ragged launch per batch vs the per-image loop and vs the equal-size stream step (also_main_batched_ragged)",
    "u8": "the uint8 integer cv2 resample alone (what the reference's drivers run: new_method.py:268-271) with its own roofline block, and 683- against 684-pixel-wide rows (also_u8)",
    "pool_input": "configs[2]'s other input form: full-resolution attention [B,1,S,S] -> 24x24 pool -> maps -> warp (also_pool_input)",
    "config5": "configs[4] data flow with a random-weight CLIP ViT-L/14-336 tower, per-leg ms (also_config5)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3,
                    help="W untimed warm-up steps in front of the K timed steps (the contract's warm-up; `--prewarm` can add more, "
                         "none by default)")
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="EXTRA untimed device pre-conditioning in front of the W warm-up steps: steps of the 1024 workload / tenths "
                         "of a second of graph replays for the 336 workloads.  Default: none (the untimed work in front of the K timed "
                         "steps of the main line is the pipeline priming + W; the three-launch `also_eager` measurement, itself W + K "
                         "steps, runs before it).  The first ~20 launches of a process run 1-2 %% slower (clock ramp): a run that times "
                         "a 336 workload as its FIRST work can add `--prewarm 5`; reported as `prewarm_steps`")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["config5", "main_batched", "main_batched_ragged"], default="1024")
    ap.add_argument("--mode", choices=["cv2", "exact"], default="cv2", help="resample arithmetic of the main line")
    ap.add_argument("--layout", choices=["hwc", "chw"], default="hwc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip every secondary measurement (the also_* legs)")
    ap.add_argument("--legs", default="all",
                    help="comma-separated secondary legs to run beside the main line (see --list-legs); 'all' (default) or 'none'")
    ap.add_argument("--list-legs", action="store_true", help="print the names of the secondary legs and exit")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="force the GPU index (smoke-testing the N>1 path on a one-GPU box with --dist-backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="build the process group even with --gpus 1 (a one-rank RCCL communicator): the run then goes through "
                         "exactly the code of a rank of N -- RCCL init, weight broadcast, all_gather / all_reduce / barrier")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: rendezvous, weight broadcast, counters and the JSON line only (CPU tests)")
    args = ap.parse_args()
    if args.list_legs:
        for name, doc in LEGS.items():
            print(f"{name:28s} {doc}")
        return
    want = set(LEGS) if args.legs == "all" else set() if args.legs == "none" else set(args.legs.split(","))
    if want - set(LEGS):
        raise SystemExit(f"bench.py: unknown leg(s) {sorted(want - set(LEGS))}; --list-legs names them")
    if args.no_also:
        want = set()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))          # parent: no torch import, no GPU call

    if args.workload in ("config5", "main_batched", "main_batched_ragged"):
        import torch
        from attwarp_amd import _lib, pipeline
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
        _lib.load()
        dev = torch.device("cuda", args.device or 0)
        torch.cuda.set_device(dev)
        res = leg_config5(dev, torch, pipeline, 64, max(args.steps // 4, 3)) if args.workload == "config5" else \
            leg_main_batched(dev, torch, pipeline, max(args.steps, 48)) if args.workload == "main_batched" else \
            leg_main_batched_ragged(dev, torch, pipeline, max(args.steps, 48))
        print(json.dumps(dict({"metric": "warped images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                               "dtype": "u8" if args.workload.startswith("main_batched") else "f16 tower / f32 MarginalNet / u8 warp",
                               "config": {"workload": res["workload"]}}, **res)), flush=True)
        return
    B, S, cfg_idx = WORKLOADS[args.workload]
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank_env = int(os.environ.get("RANK", "0"))
    if world_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}; start it as "
                         f"`python bench.py --gpus {args.gpus}` or under torch.distributed.run with --nproc-per-node {args.gpus}")
    if os.environ.get("ATTWARP_BENCH_FAIL_RANK") == str(rank_env):   # test hook of the launcher (tests/test_dist_gloo.py)
        raise SystemExit(3)

    # all-core CPU baseline: forked workers, so it runs BEFORE anything initialises the GPU in this process
    cpu_all = None
    want_cpu = rank_env == 0 and world_env == 1 and not args.no_cpu_baseline and not args.dry_run
    if want_cpu:
        cpu_all = cpu_baseline_all_cores(S, args.mode)
    # one rank per GPU: stay on the cores next to that GPU (before torch starts its threads)
    affinity = pin_to_local_cores(int(os.environ.get("LOCAL_RANK", "0")) if args.device is None else args.device,
                                  world_env) if world_env > 1 else None

    import numpy as np
    import torch
    from attwarp_amd import dist as D, _lib
    rank, world, local = D.init(args.dist_backend if not args.dry_run else (args.dist_backend or "gloo"), args.device,
                                use_gpu=not args.dry_run, force_group=args.force_dist)
    grouped = world > 1 or args.force_dist
    if args.device is not None:
        local = args.device
    _lib.load()
    ranks_seen = [int(v) for v in D.all_gather_counters({"rank": float(rank)})["rank"]]
    # a line labelled n_gpus = N must come from N ranks that all see each other, over RCCL on a GPU run: anything else exits
    # non-zero (the launcher / torchrun then ends the other ranks) instead of printing a number
    if ranks_seen != list(range(world)):
        raise SystemExit(f"bench.py: the process group gathered ranks {ranks_seen}, expected {list(range(world))}")
    if grouped and not args.dry_run:
        backend = torch.distributed.get_backend()
        if backend != "nccl" and args.dist_backend is None:
            raise SystemExit(f"bench.py: GPU run over backend {backend!r}; RCCL ('nccl') expected (pass --dist-backend explicitly "
                             "to smoke-test another one)")

    if args.dry_run:
        from attwarp_amd.model import MarginalNet
        net = MarginalNet(16, 24, 8)
        nbytes = D.broadcast_module_weights(net, src=0) if grouped else 0
        D.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))
        wall_local = time.perf_counter() - t0
        wall = D.max_over_ranks(wall_local)
        per = D.all_gather_counters({"images": float(B * args.steps), "wall_s": wall_local})
        line = {"metric": "warped images/sec", "value": None, "unit": "images/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": "weak",
                "weights_broadcast": {"bytes": nbytes}, "per_rank_images": per["images"],
                "ranks_seen": ranks_seen, "rccl_ranks_seen": ranks_seen, "cpu_affinity_rank0": affinity,
                "max_wall_s": wall, "config": {"workload": args.workload, "batch_per_gpu": B},
                "scaling_curve": SCALING_CURVE_NOTE}
        # the secondary legs every rank of a group runs behind the main line (same collectives, same keys, no GPU work):
        # agree-on-eligibility, barrier-bracketed timed region, max over ranks, gathered per-rank rates
        for key, units in dist_leg_plan(args.workload, want, grouped):
            ok = agree(D, True)
            D.barrier()
            t0 = time.perf_counter()
            time.sleep(0.005 * (1 + rank))
            D.barrier()
            wl = time.perf_counter() - t0
            line[key] = dict(rank_fields(D, world, float(units * args.steps), wl, D.max_over_ranks(wl), ok), dry_run=True)
        if rank == 0:
            print(json.dumps(line), flush=True)
        D.shutdown()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from attwarp_amd import pipeline

    # start-up collective of the multi-GPU path: one RCCL broadcast of MarginalNet weights (untimed)
    bcast = None
    if grouped:
        from attwarp_amd.model import MarginalNet
        net = MarginalNet(1024, 4096, 256).to(dev)
        torch.cuda.synchronize(); D.barrier()
        t0 = time.perf_counter()
        nbytes = D.broadcast_module_weights(net, src=0)
        torch.cuda.synchronize()
        bcast = {"bytes": nbytes, "ms": round((time.perf_counter() - t0) * 1e3, 3)}
        del net

    mode_txt = ("cv2.remap arithmetic: 1/32-px coordinates, 4 table weights" if args.mode == "cv2"
                else "unquantised bilinear = grid_sample")
    workload_txt = (f"batch-{B} {S}x{S}x3 float32 {args.layout.upper()} images per GPU + attention rows "
                    f"[T={T_STEPS},B,{HEADS},{KV}] float32 -> reduce -> 24x24 -> marginals -> CDF -> inverse maps -> "
                    f"bilinear warp, mode={args.mode} ({mode_txt}) (BASELINE configs[{cfg_idx}])")
    small = args.workload != "1024"
    extra = {}
    if small:
        # configs[1] / configs[3]: the step is ~0.09-0.25 ms, so the timed loop is a HIP-graph replay with no host call
        # per kernel (the eager three-launch line with HIP events is attached as "also_eager")
        res, wall, wall_local, step, ow = small_workload(B, S, dev, 1234 + rank, args.mode, args.layout, args.steps,
                                                         args.warmup, D, torch, pipeline,
                                                         prewarm_s=PREWARM_SMALL_S if args.prewarm < 0 else args.prewarm / 10.0)
        roof = res["eager"]["roofline"]
        roof["measured_in"] = "the eager pass right after the timed graph region (a graph replay has no per-kernel events)"
        extra = {"step_algorithmic_bytes": res["step_algorithmic_bytes"], "step_TBps": res["step_TBps"],
                 "step_frac_of_hbm_peak": res["step_frac_of_hbm_peak"], "bit_identical_to_serial": res["bit_identical_to_serial"],
                 "path": res["path"], "also_eager": res["eager"]}
        nrot = res["rotating_batches"]
        del ow
    else:
        step = Step(B, S, dev, seed=1234 + rank, mode=args.mode, layout=args.layout)
        # optional extra pre-conditioning (--prewarm; none by default: the contract's W warm-up steps are the warm-up)
        n_prewarm = PREWARM_STEPS if args.prewarm < 0 else args.prewarm
        for _ in range(n_prewarm):
            step()
        torch.cuda.synchronize()
        # (a) the plain step: reduce -> maps -> resample of ONE batch, three eager launches (reported as "also_eager")
        wall_e, _ = time_steps(step, args.steps, args.warmup, D)
        st_e = step.stage_ms()
        eager = {"workload": "the same step as three eager launches on one batch (reduce -> maps -> resample), HIP events between them",
                 "value": round(world * B * args.steps / wall_e, 1), "unit": "images/s",
                 "ms_per_step": round(wall_e / args.steps * 1e3, 4),
                 "stages_ms": {"attn_reduce_step_kernel": round(st_e[0], 4), "axis_maps_from_steps_kernel": round(st_e[1], 4),
                               "remap_rows_kernel": round(st_e[2], 4)},
                 "roofline": roofline_of(step)}
        # (b) the main line: the same work per step as a batch STREAM -- reduce of batch k+2 + maps of batch k+1 in one
        # launch, then the resample of batch k (pipeline.OverlappedWarp pattern "am", driven eagerly here so that HIP
        # events bracket the resample): one launch boundary less, and the resample no longer starts behind the
        # latency-bound 25 us map kernel (it measures ~2 % faster there)
        main_step = PipelinedStep(step)
        for _ in range(5):
            main_step()
        wall, wall_local = time_steps(main_step, args.steps, args.warmup, D)
        roof = roofline_of(main_step, load_pmc_traffic(args.workload, args.mode))
        if roof["traffic"] is not None:                 # a committed constant, not a counter read in this run
            roof["traffic_source"] = pmc_traffic_source(args.workload, args.mode)
        ref_main = pipeline.warp_from_attention_stack(step.sets[0][0], step.sets[0][1], step.starts,
                                                      channels_last=(args.layout == "hwc"), mode=args.mode)
        extra = {"bit_identical_to_serial": bool(torch.equal(main_step.ow.out, ref_main)),
                 "path": "a stream of batches, two launches per step: attention reduce of batch k+2 + map construction of batch "
                         "k+1 as ONE launch (attwarp_attn_reduce_and_maps), then the resample of batch k "
                         "(pipeline.OverlappedWarp pattern 'am', eager, HIP events around the resample); every step runs exactly "
                         "one reduce, one map construction and one resample",
                 "also_eager": eager}
        del ref_main
        nrot = step.nrot
    roof["calibration"] = calibration_of(step)
    per_rank = D.all_gather_counters({"images_per_s": B * args.steps / wall_local, "roofline_frac": roof["frac"]})
    ms_per_step = wall / args.steps * 1e3
    value = world * B * args.steps / wall

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
        "prewarm_steps": getattr(time_overlapped, "prewarm_steps", 0) if small else n_prewarm,
        "config": {"workload": workload_txt, "mode": args.mode, "batch_per_gpu": B, "image_size": S,
                   "layout": args.layout.upper(), "global_batch": world * B, "rotating_batches": nrot,
                   "sharding": "contiguous image blocks per rank, no data-path collective"},
        "roofline": roof,
    }
    result.update(extra)
    if small:
        st_ms = step.stage_ms()
        result["stages_ms"] = {"attn_reduce_step_kernel": round(st_ms[0], 4), "axis_maps_from_steps_kernel": round(st_ms[1], 4),
                               "remap_rows_kernel": round(st_ms[2], 4)}
    else:
        st_ms = main_step.stage_ms()
        result["stages_ms"] = {"attn_maps_kernel": round(st_ms[0], 4), "remap_rows_kernel": round(st_ms[1], 4)}
        del main_step
        torch.cuda.empty_cache()
    if grouped:
        rates = per_rank["images_per_s"]
        result["per_rank_images_per_s"] = [round(v, 1) for v in rates]
        result["per_rank_roofline_frac"] = [round(v, 4) for v in per_rank["roofline_frac"]]
        # `value` is N * B * K / the SLOWEST rank's time, and GPUs of one node differ by several per cent: this
        # is the job's rate against N x the mean rank -- 1.0 means the ranks lose nothing to each other beyond their own
        # spread -- so the curve can be read apart from which GPU the N=1 run happened to land on
        result["scaling_efficiency_vs_rank_mean"] = round(value / (world * sum(rates) / len(rates)), 4)
        result["rccl_ranks_seen"] = ranks_seen
        result["dist_backend"] = torch.distributed.get_backend()
        result["cpu_affinity_rank0"] = affinity
    if bcast:
        result["weights_broadcast"] = bcast

    if want_cpu:
        n_max = min(B, 16 if S >= 1024 else 64)
        img = step.img[:n_max].cpu().numpy()
        if args.layout == "chw":
            img = np.ascontiguousarray(img.transpose(0, 2, 3, 1))
        n, dt, out0 = _cpu_loop(img, step.rows[:, :n_max].cpu().numpy(), step.starts[:n_max].cpu().numpy(), S, args.mode, 10.0)
        g0 = pipeline.warp_from_attention_stack(step.img[:1], step.rows[:, :1].contiguous(), step.starts[:1],
                                                channels_last=(args.layout == "hwc"), mode=args.mode)[0].cpu().numpy()
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

    one_gpu_1024 = world == 1 and not small
    if one_gpu_1024:
        ctx = dict(step=step, args=args, D=D, torch=torch, pipeline=pipeline, B=B)
        if "exact" in want:
            result.update(leg_other_mode(**ctx))
        if "chw" in want:
            result.update(leg_other_layout(**ctx))
        if "fused" in want:
            result.update(leg_fused_1024(**ctx))
        if "distributions" in want:
            result.update(leg_distributions(**ctx))
    del step
    torch.cuda.empty_cache()

    n2 = max(args.steps, 48)
    result["scaling_curve"] = SCALING_CURVE_NOTE
    rk = dict(rank=rank, world=world, grouped=grouped)
    if world > 1:
        # every rank of the group runs the legs of dist_leg_plan on its own batches (no try around them: a failure inside a
        # timed collective region must end the job -- the launcher / torchrun kills the other ranks -- not hang it)
        for key, _ in dist_leg_plan(args.workload, want, grouped):
            if key == "also_336x256":
                result[key] = leg_small("336x256", dev, args, n2, D, torch, pipeline, **rk)
            else:
                result[key] = leg_main_batched_ragged(dev, torch, pipeline, n2, D, **rk)
            torch.cuda.empty_cache()
    if world == 1 and args.workload == "1024" and "336" in want:
        for key, wl in (("also", "336"), ("also_336x256", "336x256")):
            result[key] = leg_small(wl, dev, args, n2, D, torch, pipeline, **rk)
    if world == 1 and args.workload == "1024" and "fp16_attention" in want:
        # configs[3]'s per-rank batch with the attention rows in float16, the dtype LLaVA-1.5 emits (half the reduce's bytes)
        result["also_336x256_fp16_attention"] = leg_small("336x256", dev, args, n2, D, torch, pipeline, attn_dtype=torch.float16)
    if world == 1 and args.workload == "1024":
        # (a failure inside one of these legs -- e.g. `transformers` missing for the vision tower -- must not cost the main line)
        for key, leg in (("main_batched", lambda: leg_main_batched(dev, torch, pipeline, n2)),
                         ("main_batched_ragged", lambda: leg_main_batched_ragged(dev, torch, pipeline, n2, D, **rk)),
                         ("u8", lambda: leg_u8(dev, torch, pipeline, args.steps)),
                         ("pool_input", lambda: leg_pool_input(dev, torch, pipeline, B, S, args.mode, args.steps)),
                         ("config5", lambda: leg_config5(dev, torch, pipeline, 32, 3))):
            if key in want:
                try:
                    result[f"also_{key}"] = leg()
                except Exception as e:      # noqa: BLE001
                    result[f"also_{key}"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                torch.cuda.empty_cache()

    if rank == 0:
        print(json.dumps(result), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
