"""Dev tool: rows-per-block x row-blocks-per-workgroup sweep of the integer uint8 cv2 resample."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline, checkpoint_utils as cu, _lib
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
ONLY1024 = "only1024" in sys.argv
for (B, S, So, layout, kind) in [(256, 1024, 1024, "hwc", "uniform"), (256, 1024, 1024, "hwc", "peaked"), (256, 1024, 500, "hwc", "peaked"),
                                 (256, 336, 500, "hwc", "peaked"), (64, 336, 500, "hwc", "peaked"), (256, 1024, 1024, "chw", "uniform")]:
    if ONLY1024 and not (S == 1024 and So == 1024 and layout == "hwc"): continue
    shape = (B, S, S, 3) if layout == "hwc" else (B, 3, S, S)
    img8 = (torch.rand(*shape, device=dev) * 255).to(torch.uint8)
    px = torch.softmax(torch.randn(B, 24, device=dev) * (0.02 if kind == "uniform" else 1.0), 1)
    mx, my = pipeline.axis_maps_from_pdf(px, px, (S, S), (So, So))
    ref = cu.remap_separable(img8, mx, my, mode="cv2", channels_last=(layout == "hwc"))
    res = []
    G = [int(a[1:]) for a in sys.argv[1:] if a.startswith("g")] or [-1]
    for R, c, g in [(R, c, g) for g in G for (R, c) in [(-1, -1), (32, 2), (16, 2), (16, 4), (8, 2), (8, 4), (8, 8), (4, 4), (4, 8), (6, 4), (12, 2)]]:
        with _lib.debug_override(remap_rows=R, remap_cpw=c, remap_noswz=g):
            out = cu.remap_separable(img8, mx, my, mode="cv2", channels_last=(layout == "hwc"))
            assert torch.equal(out, ref), (R, c)
            ms = t(lambda: cu.remap_separable(img8, mx, my, mode="cv2", channels_last=(layout == "hwc"), out=out))
        res.append(f"R{R}c{c}g{g} {ms*1e3:.1f}")
    print(f"u8 cv2 {layout} {kind} B={B} {S}->{So}: " + "  ".join(res) + f"   [{B*(S*S*3+So*So*3)/1e6:.0f} MB]", flush=True)
