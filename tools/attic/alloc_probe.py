"""Dev tool: what moves the float32 resample between its "fast" and "slow" states on ONE box, in ONE process?
 (1) buffers carved from one arena vs separate torch allocations, (2) the distance between the eight per-XCD streams
 (batch size), (3) a per-XCD start skew inside the contiguous ranges.   python tools/alloc_probe.py"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import checkpoint_utils as cu, _lib
from remap_bench import maps

S = 1024
dev = torch.device("cuda:0")
IMG = S * S * 3 * 4
mxs = {}

def timeit(fn, k=8):
    ts = []
    for _ in range(k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)

def line(tag, src, dst, B, variants):
    if B not in mxs:
        mxs[B] = maps(B, S, "uniform")
    mx, my = mxs[B]
    res = []
    for name, over in variants:
        with _lib.debug_override(**over):
            t = timeit(lambda: cu.remap_separable(src, mx, my, mode="cv2", channels_last=True, out=dst))
        res.append(f"{name} {t * 256 / B:.4f}")
    t = timeit(lambda: torch.add(src, 1.0, out=dst))
    print(f"{tag:34s} " + "  ".join(res) + f"  add {t * 256 / B:.4f}   (ms per 256 images)", flush=True)

BASIC = [("default", {}), ("R3g4", dict(remap_rows=3, remap_noswz=4)), ("plain", dict(remap_noswz=1)),
         ("R4c2", dict(remap_cpw=2))]
SKEWS = [("default", {})] + [(f"skew{k}", dict(remap_skew=k)) for k in (1, 3, 7, 16, 37, 64, 129, 511, 1000)]
SKEWS2 = [("c2", dict(remap_cpw=2))] + [(f"c2skew{k}", dict(remap_cpw=2, remap_skew=k)) for k in (3, 16, 37, 129, 1000)]

# (a) separate torch allocations, three fresh pairs
pairs = []
for i in range(2):
    s2 = torch.rand((256, S, S, 3), device=dev); d2 = torch.empty_like(s2)
    pairs.append((s2, d2))
    line(f"torch pair {i} src {s2.data_ptr() >> 20:#x}M", s2, d2, 256, BASIC)
line("torch pair 0 again", *pairs[0], 256, BASIC)
line("torch pair 0 skews", *pairs[0], 256, SKEWS)
line("torch pair 0 skews cpw2", *pairs[0], 256, SKEWS2)
line("torch src0 -> dst1", pairs[0][0], pairs[1][1], 256, BASIC)
# (b) the spacing of the per-XCD streams: smaller batches out of the same buffers
for B in (255, 250, 248, 240, 224, 200, 192, 128):
    line(f"torch pair 0, B={B}", pairs[0][0][:B], pairs[0][1][:B], B, BASIC)
del pairs, s2, d2
torch.cuda.empty_cache()
# (c) one arena, src and dst carved out of it
nbytes = 256 * IMG
arena = torch.empty(2 * nbytes + (1 << 30), dtype=torch.uint8, device=dev)
fa = arena[: (arena.numel() // 4) * 4].view(torch.float32)
for i in range(0, fa.numel(), 1 << 28):
    fa[i:i + (1 << 28)].uniform_(0, 1)
def view(off, B=256):
    return arena[off:off + B * IMG].view(torch.float32).view(B, S, S, 3)
line(f"arena {arena.data_ptr() >> 20:#x}M src@0 dst@+N+64K", view(0), view(nbytes + 65536), 256, BASIC)
line("arena skews", view(0), view(nbytes + 65536), 256, SKEWS)
line("arena skews cpw2", view(0), view(nbytes + 65536), 256, SKEWS2)
for B in (250, 240, 192):
    line(f"arena B={B}", view(0, B), view(nbytes + 65536, B), B, BASIC)
line("arena again", view(0), view(nbytes + 65536), 256, BASIC)
