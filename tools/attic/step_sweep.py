"""Dev tool (round 4): rows per block x row blocks per workgroup of the resample INSIDE the one-launch 336x336 step
(pipeline.OverlappedWarp, pattern "fused"; graphs are captured under the attwarp_debug_set overrides), one process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline, _lib
dev = torch.device("cuda:0")
K = 96
variants = [dict()] + [dict(remap_rows=R, remap_cpw=c) for R in (4, 6, 8, 12) for c in (1, 2, 4) if R * c <= 48]
for B, S in ((64, 336), (256, 336)):
    batch_bytes = 2 * B * S * S * 3 * 4 + 20 * B * 32 * 640 * 4
    n = max(4, min(8, -(-(2 << 30) // batch_bytes))); n += n % 2
    g = torch.Generator(device=dev).manual_seed(B)
    imgs = [torch.rand(B, S, S, 3, device=dev, generator=g) for _ in range(n)]
    rows = [torch.softmax(torch.randn(20, B, 32, 640, device=dev, generator=g), -1) for _ in range(n)]
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    res = {}
    for rep in range(2):
        for over in variants:
            with _lib.debug_override(**over):
                ow = pipeline.OverlappedWarp(imgs, rows, starts, channels_last=True, pattern="fused")
                def run():
                    ow.reset(); ow.prime(); ow.prime2(); ow.run(K - 2); ow.tail()
                run(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / K)
            res.setdefault(tuple(sorted(over.items())), []).append(best * 1e6)
            del ow
    for k, v in sorted(res.items(), key=lambda kv: min(kv[1])):
        print(f"B={B} {S}x{S} {dict(k) or 'default'}: " + " ".join(f"{x:.1f}" for x in v) + " us per step", flush=True)
    del imgs, rows
    torch.cuda.empty_cache()
