import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
x = torch.empty(1 << 28, device=dev)
for _ in range(200): x.add_(1.0)
for (B, S, dt) in [(64, 336, torch.uint8), (256, 336, torch.uint8), (64, 1024, torch.uint8), (256, 336, torch.float16)]:
    n = 8 if B == 64 and S == 336 else 2
    g = torch.Generator(device=dev).manual_seed(1)
    if dt == torch.uint8:
        imgs = [(torch.rand((B, S, S, 3), device=dev, generator=g) * 255).to(torch.uint8) for _ in range(n)]
    else:
        imgs = [torch.rand((B, S, S, 3), device=dev, generator=g) for _ in range(n)]
    rws = [torch.softmax(torch.randn((20, B, 32, 640), device=dev, generator=g), -1).to(torch.float16 if dt == torch.float16 else torch.float32) for _ in range(n)]
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    for pat in ("dag", "am", "join") + (("fused",) if dt != torch.uint8 else ()):
        ow = pipeline.OverlappedWarp(imgs, rws, starts, channels_last=True, pattern=pat)
        ow.prime(); ow.prime2(); ow.run(16); torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); ow.run(64); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 64)
        print(f"B={B} S={S} img={dt} pattern={pat}: {best*1e3:.4f} ms/step", flush=True)
        del ow
