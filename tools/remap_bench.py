"""Dev tool: time the remap kernel alone on the BASELINE sizes with near-identity and peaked maps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import checkpoint_utils as cu, pipeline

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

def bench(B, S, layout, kind, iters=20):
    img = torch.rand((B, S, S, 3) if layout == "hwc" else (B, 3, S, S), device=dev)
    out = torch.empty_like(img)
    mx, my = maps(B, S, kind)
    for _ in range(3): cu.remap_separable(img, mx, my, channels_last=(layout == "hwc"), out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); cu.remap_separable(img, mx, my, channels_last=(layout == "hwc"), out=out); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts)//2]
    gb = 2 * B * S * S * 3 * 4 / 1e9
    print(f"B={B:4d} S={S:4d} {layout} {kind:8s} R={os.environ.get('ATTWARP_REMAP_ROWS','auto'):>4s} var={os.environ.get('ATTWARP_REMAP_VARIANT','r')}: "
          f"{ms:.4f} ms (min {min(ts):.4f})  {gb/ms:.2f} TB/s  {100*gb/ms/8:.1f}% of 8TB/s")

def run(env, *args):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    print({k.replace("ATTWARP_REMAP_", ""): v for k, v in env.items()}, end="  ")
    bench(*args)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v

if __name__ == "__main__":
    for layout in ("hwc", "chw"):
        for kind in ("uniform", "peaked"):
            bench(256, 1024, layout, kind)
    bench(64, 336, "hwc", "uniform", 50)
    bench(256, 336, "hwc", "uniform", 50)
    bench(256, 336, "chw", "uniform", 50)
    run({"ATTWARP_REMAP_VARIANT": "g"}, 256, 1024, "hwc", "uniform")
