"""Dev tool: the 336x336 stream step as one launch per batch (pipeline.OverlappedWarp, pattern "fused") against one launch
per TWO batches (pipeline.PairedStepWarp), same ring of independent batches (>= 2 GiB), same process, alternating."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
K = 96
for B, S in ((64, 336), (256, 336)):
    batch_bytes = 2 * B * S * S * 3 * 4 + 20 * B * 32 * 640 * 4
    n = max(4, min(8, -(-(2 << 30) // batch_bytes)))
    n += n % 2
    g = torch.Generator(device=dev).manual_seed(B)
    imgs = [torch.rand(B, S, S, 3, device=dev, generator=g) for _ in range(n)]
    rows = [torch.softmax(torch.randn(20, B, 32, 640, device=dev, generator=g), -1) for _ in range(n)]
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    for adt in (torch.float32, torch.float16):
        rws = [r.to(adt) for r in rows]
        ow = pipeline.OverlappedWarp(imgs, rws, starts, channels_last=True, pattern="fused")
        pw = pipeline.PairedStepWarp(imgs, rws, starts, channels_last=True)
        refs = [pipeline.warp_from_attention_stack(imgs[i], rws[i], starts, channels_last=True) for i in range(n)]
        def run_ow():
            ow.reset(); ow.prime(); ow.prime2(); ow.run(K - 2); ow.tail()
        def run_pw():
            pw.reset(); pw.prime(); pw.run(K - 4); pw.tail()
        run_ow(); run_pw(); torch.cuda.synchronize()
        same = all(torch.equal(pw.outs[i], refs[i]) for i in range(n)) and all(torch.equal(ow.outs[i], refs[i]) for i in range(n))
        best = {"one launch per batch": 1e9, "one launch per two batches": 1e9}
        for rep in range(6):
            for name, fn in (("one launch per batch", run_ow), ("one launch per two batches", run_pw)):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
                best[name] = min(best[name], (time.perf_counter() - t0) / K)
        for name, t in best.items():
            print(f"B={B} {S}x{S} rows {str(adt)[6:]:8s} ring {n}: {name:28s} {t*1e6:7.1f} us per batch  bit_identical={same}", flush=True)
        del ow, pw, refs, rws
    del imgs, rows
    torch.cuda.empty_cache()
