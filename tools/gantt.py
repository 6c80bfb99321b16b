"""Dev tool: block timeline of the one-launch steps (tuning flavour only: common.hpp trace_block writes
{start, end (100 MHz clock), kind, HW_ID | XCC_ID << 32} per block into a buffer handed over through attwarp_debug_set).
usage: python tools/gantt.py chain B S So [tune=key:value,...]      mask_chain_step_kernel   (kinds F V P L R)
       python tools/gantt.py step B S [f16] [tune=...]              warp_step_kernel         (kinds M A R)
       python tools/gantt.py remap B S [exact] [tune=...]           remap_rows_kernel        (the float32 resample alone)
       python tools/gantt.py ragged B [tune=...]                    mask_chain_ragged_kernel (TextVQA-like size mix, -> 500 x 500)
Prints, for the LAST step of a graph-replayed stream in its steady state: when each kind of block starts and ends, how long
its blocks take, how many blocks are resident over time (2 us bins) and which kind holds the tail of the launch."""
import os, sys, time, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline, _lib
if os.environ.get("AB_TUNING_LIB"):          # another build of the tuning flavour (same record layout)
    _lib.TUNING_LIB_PATH = os.path.abspath(os.environ["AB_TUNING_LIB"])
dev = torch.device("cuda:0")
pos = [a for a in sys.argv[1:] if "=" not in a]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
over = {k: int(v) for k, v in (kv.split(":") for kv in kw.get("tune", "").split(",") if kv)}
what = pos[0]
NREC = 1 << 21
W = 16                                   # common.hpp: TRACE_WORDS
trace = torch.zeros(NREC * W, dtype=torch.int64, device=dev)
addr = trace.data_ptr()
assert addr < (1 << 48)
over.update(trace_lo=addr & 0xFFFFFF, trace_hi=addr >> 24)
K = int(kw.get("steps", "48"))

if what == "chain":
    B, S, So = int(pos[1]), int(pos[2]), int(pos[3])
    names = {0: "F finalize", 1: "V revise", 2: "P marginals", 3: "L lanczos", 4: "R resample", -1: "(padding)"}
    slot = B * (3 * S * S + 3 * So * So)
    n = max(2, min(16, -(-(1 << 30) // slot))); n += n & 1
    g = torch.Generator(device=dev).manual_seed(B + S)
    images = [torch.randint(0, 256, (B, S, S, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    masks = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
    with _lib.debug_override(**over):
        mc = pipeline.MaskChainStream(images, masks, (So, So), pattern="fused")
        def run(steps):
            mc.reset(); mc.prime(); mc.run(steps); mc.drain()
        run(K); torch.cuda.synchronize()
    depth = mc.depth
    run_steady = lambda: mc.run(K)
elif what == "ragged":
    B = int(pos[1])
    names = {0: "F finalize", 1: "V revise", 2: "P marginals", 3: "L lanczos", 4: "R resample", -1: "(padding)"}
    WH = [(1024, 768), (683, 1024), (1024, 1024), (500, 375), (333, 500), (640, 427)]
    sizes = [WH[b % len(WH)] for b in range(B)]
    px = sum(w * h for (w, h) in sizes)
    n = max(6, min(24, -(-(1 << 30) // (3 * px + 3 * B * 250000))))
    g = torch.Generator(device=dev).manual_seed(B)
    ring = []
    ctx = _lib.debug_override(**over)
    ctx.__enter__()          # the tables are planned under the overrides too; eager launches: they stay on for the whole run
    for _ in range(n):
        rb = pipeline.RaggedBatch([torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (w, h) in sizes], (500, 500))
        rb.masks = torch.rand(B, 24, 24, device=dev, generator=g)
        ring.append(rb)
    st_ = pipeline.RaggedMaskChainStream(out_size=(500, 500))
    st_.ring(ring)
    depth = 0
    st_.prime()
    def run_steady():
        st_.run(K)
    run_steady(); torch.cuda.synchronize()
elif what == "remap":
    from attwarp_amd import checkpoint_utils as cu
    B, S = int(pos[1]), int(pos[2])
    mode = "exact" if "exact" in pos else "cv2"
    names = {2: "R resample", -1: "(padding)"}
    n = max(2, -(-(2 << 30) // (2 * B * S * S * 3 * 4)))
    g = torch.Generator(device=dev).manual_seed(B)
    imgs = [torch.rand(B, S, S, 3, device=dev, generator=g) for _ in range(n)]
    outs = [torch.empty_like(imgs[0]) for _ in range(n)]
    px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.02, 1)     # near identity, as averaged random attention gives
    py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.02, 1)
    mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
    depth = 0
    ctx = _lib.debug_override(**over)
    ctx.__enter__()          # eager launches: the overrides stay on for the whole run
    def run_steady():
        for i in range(K):
            cu.remap_separable(imgs[i % n], mx, my, mode=mode, channels_last=True, out=outs[i % n])
    run_steady(); torch.cuda.synchronize()
else:
    B, S = int(pos[1]), int(pos[2])
    adt = torch.float16 if "f16" in pos else torch.float32
    names = {0: "M maps", 1: "A reduce", 2: "R resample", -1: "(padding)"}
    batch_bytes = 2 * B * S * S * 3 * 4 + 20 * B * 32 * 640 * 4
    n = max(4, min(8, -(-(2 << 30) // batch_bytes))); n += n % 2
    g = torch.Generator(device=dev).manual_seed(B)
    imgs = [torch.rand(B, S, S, 3, device=dev, generator=g) for _ in range(n)]
    rows = [torch.softmax(torch.randn(20, B, 32, 640, device=dev, generator=g), -1).to(adt) for _ in range(n)]
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    with _lib.debug_override(**over):
        ow = pipeline.OverlappedWarp(imgs, rows, starts, channels_last=True, pattern="fused")
        def run(steps):
            ow.reset(); ow.prime(); ow.prime2(); ow.run(steps - 2); ow.tail()
        run(K); torch.cuda.synchronize()
    depth = 0
    run_steady = lambda: ow.run(K)

# graphs were captured under the overrides: the trace pointer is part of their kernel arguments.  Time the steady state,
# then read the records of the last full step (drain / tail launches write too: read before them).
best = 1e9
for rep in range(5):
    if what == "chain":
        mc.reset(); mc.prime()
    elif what == "step":
        ow.reset(); ow.prime(); ow.prime2()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_steady()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K)
rec = trace.view(-1, W).cpu().numpy()
rec = rec[rec[:, 1] != 0]
t0 = rec[:, 0].min()
st = (rec[:, 0] - t0) / 100.0          # us
en = (rec[:, 1] - t0) / 100.0
kind = rec[:, 2]
xcc = (rec[:, 3] >> 32) & 0xF
span = en.max()
print(f"{what} {' '.join(pos[1:])} {kw.get('tune', '')}: {best*1e6:.1f} us per step (stream, best of 5 x {K}); last step: {len(rec)} blocks, "
      f"first start -> last end {span:.1f} us  (step - span = {best*1e6 - span:.1f} us between launches)")
print(f"{'kind':14s} {'blocks':>7s} {'first start':>11s} {'last start':>10s} {'first end':>9s} {'last end':>8s} {'mean dur':>8s} {'max dur':>7s}  block-us")
for k in sorted(set(kind.tolist())):
    m = kind == k
    d = en[m] - st[m]
    print(f"{names.get(int(k), str(k)):14s} {m.sum():7d} {st[m].min():11.1f} {st[m].max():10.1f} {en[m].min():9.1f} {en[m].max():8.1f} {d.mean():8.2f} {d.max():7.2f}  {d.sum():9.0f}")
# in-kernel shader clock: d s_memtime / d s_memrealtime x 100 MHz, median over blocks of >= 5 us
long_ = (rec[:, 1] - rec[:, 0]) >= 500
clk = float(np.median((rec[long_, 5] - rec[long_, 4]) / (rec[long_, 1] - rec[long_, 0]) * 0.1)) if long_.any() else float("nan")
print(f"in-kernel shader clock {clk:.2f} GHz (median over {int(long_.sum())} blocks)")
# a body's marks (shader-clock stamps): mean cycles between successive marks, per kind
for k in sorted(set(kind.tolist())):
    m = (kind == k) & (rec[:, 6] == 1)
    if not m.any():
        continue
    stamps = rec[m][:, 7:16].astype(np.float64)
    nz = int((stamps[0] != 0).sum())
    prev = rec[m, 4].astype(np.float64)
    parts = []
    for i in range(nz):
        parts.append(f"->m{i} {(stamps[:, i] - prev).mean() / clk / 1e3:.2f}")
        prev = stamps[:, i]
    parts.append(f"->end {(rec[m, 5] - prev).mean() / clk / 1e3:.2f}")
    print(f"{names.get(int(k), str(k))}: us between marks (start {' '.join(parts)})")
# phase counters of the resample blocks (shader-clock cycles summed over the block's rows)
m = rec[:, 6] == 2
if m.any():
    rows = rec[m, 10].astype(np.float64)
    dur = (en[m] - st[m])
    cyc = (rec[m, 5] - rec[m, 4]).astype(np.float64)
    ph = [rec[m, 7 + i] / rows for i in range(3)]
    mk = rec[m][:, 11:15].astype(np.float64)
    if (mk[:, 0] > 0).all():
        c0 = rec[m, 4].astype(np.float64)
        print(f"resample blocks, cycles from block start: column taps done {(mk[:, 0] - c0).mean():.0f}, row maps in LDS {(mk[:, 1] - c0).mean():.0f}, "
              f"first rows requested {(mk[:, 2] - c0).mean():.0f}, last stores issued {(mk[:, 3] - c0).mean():.0f}, end {cyc.mean():.0f}")
    print(f"resample blocks, cycles per output row: load wait + staging {ph[0].mean():.0f}, look-ahead issue + barrier {ph[1].mean():.0f}, "
          f"gather + arithmetic + stores {ph[2].mean():.0f}; outside the row loop (taps prologue) {((cyc - rec[m, 7] - rec[m, 8] - rec[m, 9]) / rows).mean():.0f}; "
          f"rows per block {rows.mean():.1f}, block time per row {(dur / rows).mean():.2f} us")
work = kind >= 0
binw = float(kw.get("bin", "2"))
nb = int(np.ceil(span / binw))
print(f"resident blocks per {binw:g} us bin (mean over the bin), by kind:")
ks = [k for k in sorted(set(kind.tolist())) if k >= 0]
print("   t us  " + " ".join(f"{names[int(k)].split()[0]:>6s}" for k in ks) + "   total")
for i in range(nb):
    lo, hi = i * binw, (i + 1) * binw
    row = []
    for k in ks:
        m = kind == k
        ov = np.clip(np.minimum(en[m], hi) - np.maximum(st[m], lo), 0, None).sum() / binw
        row.append(ov)
    print(f"{lo:7.1f}  " + " ".join(f"{v:6.0f}" for v in row) + f"  {sum(row):6.0f}")
per_x = [int((xcc[work] == x).sum()) for x in range(8)]
last_x = [float(en[work & (xcc == x)].max()) if per_x[x] else 0.0 for x in range(8)]
print("blocks per XCC:", per_x, " last end per XCC (us):", [round(v, 1) for v in last_x])
