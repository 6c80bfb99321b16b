"""scratch: ablations of the integer uint8 resample (tuning build, dbg bits through remap_skew)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline, checkpoint_utils as cu, _lib
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
B, S = 256, 1024
img8 = (torch.rand(B, S, S, 3, device=dev) * 255).to(torch.uint8)
for kind, sc in (("uniform", 0.02), ("random", 1.0)):
    px = torch.softmax(torch.randn(B, 24, device=dev) * sc, 1)
    mx, my = pipeline.axis_maps_from_pdf(px, px, (S, S), (S, S))
    out = torch.empty_like(img8)
    for dbg in (0, 1, 2, 3, 4, 5, 6, 7):
        with _lib.debug_override(remap_skew=dbg):
            ms = t(lambda: cu.remap_separable(img8, mx, my, mode="cv2", channels_last=True, out=out))
        print(kind, "dbg", dbg, "[noload]" if dbg & 1 else "", "[nostore]" if dbg & 2 else "", "[nogather]" if dbg & 4 else "", f"{ms*1e3:.1f} us", flush=True)
a = torch.empty(B * S * S * 3 // 4, device=dev, dtype=torch.int32); b2 = torch.empty_like(a)
print("torch.add same bytes", f"{t(lambda: torch.add(a, 1, out=b2))*1e3:.1f} us")
