"""Dev tool: time the remap kernel alone on the BASELINE sizes with near-identity and peaked maps, both arithmetic
modes.  `python tools/remap_bench.py [key=value ...]` forwards key=value pairs to attwarp_debug_set."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import checkpoint_utils as cu, pipeline, _lib

dev = torch.device("cuda:0")

def maps(B, S, kind):
    g = torch.Generator(device=dev).manual_seed(1)
    if kind == "uniform":      # what averaged random attention gives: near identity
        px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.02, 1)
        py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.02, 1)
    else:
        px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 2, 1)
        py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 2, 1)
    return pipeline.axis_maps_from_pdf(px, py, (S, S))

def bench(B, S, layout, kind, mode="exact", iters=20, dtype=torch.float32, tag="", **over):
    shape = (B, S, S, 3) if layout == "hwc" else (B, 3, S, S)
    img = torch.rand(shape, device=dev)
    if dtype == torch.uint8:
        img = (img * 255).to(torch.uint8)
    out = torch.empty_like(img)
    mx, my = maps(B, S, kind)
    with _lib.debug_override(**over):
        for _ in range(3): cu.remap_separable(img, mx, my, mode=mode, channels_last=(layout == "hwc"), out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); cu.remap_separable(img, mx, my, mode=mode, channels_last=(layout == "hwc"), out=out); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts)//2]
    gb = 2 * B * S * S * 3 * img.element_size() / 1e9
    print(f"B={B:4d} S={S:4d} {layout} {str(dtype)[6:]:7s} {mode:5s} {kind:8s} {over or ''} {tag}: "
          f"{ms:.4f} ms (min {min(ts):.4f})  {gb/ms:.2f} TB/s  {100*gb/ms/8:.1f}% of 8TB/s", flush=True)
    return ms

if __name__ == "__main__":
    over = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
    for mode in ("exact", "cv2"):
        for layout in ("hwc", "chw"):
            for kind in ("uniform", "peaked"):
                bench(256, 1024, layout, kind, mode, **over)
        bench(64, 336, "hwc", "uniform", mode, 50, **over)
        bench(256, 336, "hwc", "uniform", mode, 50, **over)
        bench(256, 336, "chw", "uniform", mode, 50, **over)
        for layout in ("hwc", "chw"):
            bench(256, 1024, layout, "uniform", mode, dtype=torch.uint8, **over)
        bench(256, 336, "hwc", "uniform", mode, 50, dtype=torch.uint8, **over)
    bench(256, 1024, "hwc", "uniform", "exact", remap_variant=1)
    bench(256, 1024, "hwc", "uniform", "cv2", remap_variant=1)
    for R in (2, 4, 8, 16):
        bench(256, 1024, "hwc", "uniform", "cv2", remap_rows=R)
