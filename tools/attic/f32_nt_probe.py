"""Dev tool: nontemporal stores / loads of the float32 resample (tuning flavour, remap_nt bits) on batches that do not fit
the Infinity Cache, and inside the one-launch stream step at 336x336."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline, checkpoint_utils as cu, _lib
dev = torch.device("cuda:0")
def t(fn, n=15):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
x = torch.empty(1 << 28, device=dev)
for _ in range(200): x.add_(1.0)
for (B, S, layout) in ((256, 1024, "hwc"), (256, 336, "hwc"), (512, 336, "hwc"), (256, 512, "hwc"), (256, 1024, "chw")):
    shape = (B, S, S, 3) if layout == "hwc" else (B, 3, S, S)
    img = torch.rand(*shape, device=dev)
    px = torch.softmax(torch.randn(B, 24, device=dev) * 0.3, 1)
    mx, my = pipeline.axis_maps_from_pdf(px, px, (S, S), (S, S))
    ref = cu.remap_separable(img, mx, my, mode="cv2", channels_last=(layout == "hwc")); out = torch.empty_like(ref)
    res = {}
    for cyc in range(3):
        for nt in (0, 2, 3):
            with _lib.debug_override(remap_nt=nt):
                cu.remap_separable(img, mx, my, mode="cv2", channels_last=(layout == "hwc"), out=out); assert torch.equal(out, ref)
                res.setdefault(nt, []).append(t(lambda: cu.remap_separable(img, mx, my, mode="cv2", channels_last=(layout == "hwc"), out=out)))
    print(f"f32 {layout} {S} B={B}: " + "  ".join(f"nt{nt} {sorted(v)[1]*1e3:.1f}" for nt, v in res.items()), flush=True)
# the one-launch stream step at 336x336 (rings >= 2 GiB)
for B, n in ((64, 8), (256, 2)):
    g = torch.Generator(device=dev).manual_seed(1)
    imgs = [torch.rand((B, 336, 336, 3), device=dev, generator=g) for _ in range(n)]
    rws = [torch.softmax(torch.randn((20, B, 32, 640), device=dev, generator=g), -1) for _ in range(n)]
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    for nt in (0, 2, 0, 2):
        with _lib.debug_override(remap_nt=nt):
            ow = pipeline.OverlappedWarp(imgs, rws, starts, channels_last=True, pattern="fused")
            ow.prime(); ow.prime2(); ow.run(32); torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter(); ow.run(128); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 128)
        print(f"fused step 336 B={B} remap_nt={nt}: {best*1e3:.4f} ms", flush=True)
        del ow
