"""Dev tool: the stream patterns of pipeline.OverlappedWarp at B=256 1024x1024 float32 (graph replay, cycled in one process)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
B, S = 256, 1024
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((B, S, S, 3), device=dev, generator=g)
rows = torch.softmax(torch.randn((20, B, 32, 640), device=dev, generator=g), -1)
starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
x = torch.empty(1 << 28, device=dev)
for _ in range(200): x.add_(1.0)
ows = {p: pipeline.OverlappedWarp(img, rows, starts, channels_last=True, pattern=p) for p in ("am", "dag", "join", "fused")}
for p, ow in ows.items():
    ow.prime(); ow.prime2(); ow.run(16)
torch.cuda.synchronize()
for cyc in range(3):
    for p, ow in ows.items():
        t0 = time.perf_counter(); ow.run(48); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 48
        print(f"cycle {cyc} pattern {p}: {dt*1e3:.4f} ms/step", flush=True)
