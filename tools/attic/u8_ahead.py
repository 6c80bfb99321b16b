"""Dev tool: integer uint8 cv2 resample, rows requested ahead (1 / 2 / 4) x (rows per block, blocks per workgroup), cycled
in one process so that every variant sees the same lease and clock state (medians over the cycles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline, checkpoint_utils as cu, _lib
dev = torch.device("cuda:0")
def t(fn, n=12):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
x = torch.empty(1 << 28, device=dev)
for _ in range(200): x.add_(1.0)          # past the first-second clock ramp
torch.cuda.synchronize()
variants = [(a, R, c) for a in (1, 2, 4) for (R, c) in [(-1, -1), (16, 4), (8, 8), (32, 1), (32, 2), (16, 1)]]
for (B, S, So, layout, kind) in [(256, 1024, 1024, "hwc", "uniform"), (256, 1024, 1024, "hwc", "random"), (256, 1024, 500, "hwc", "random"),
                                 (64, 1024, 500, "hwc", "random"), (256, 336, 500, "hwc", "random"), (64, 336, 500, "hwc", "random"),
                                 (256, 336, 336, "chw", "random"), (256, 1024, 1024, "chw", "uniform")]:
    shape = (B, S, S, 3) if layout == "hwc" else (B, 3, S, S)
    img8 = (torch.rand(*shape, device=dev) * 255).to(torch.uint8)
    px = torch.softmax(torch.randn(B, 24, device=dev) * (0.02 if kind == "uniform" else 1.0), 1)
    mx, my = pipeline.axis_maps_from_pdf(px, px, (S, S), (So, So))
    ref = cu.remap_separable(img8, mx, my, mode="cv2", channels_last=(layout == "hwc"))
    out = torch.empty_like(ref)
    res = {v: [] for v in variants}
    for cyc in range(3):
        for v in variants:
            a, R, c = v
            with _lib.debug_override(u8_ahead=a, remap_rows=R, remap_cpw=c):
                if cyc == 0:
                    out.zero_(); cu.remap_separable(img8, mx, my, mode="cv2", channels_last=(layout == "hwc"), out=out)
                    assert torch.equal(out, ref), v
                res[v].append(t(lambda: cu.remap_separable(img8, mx, my, mode="cv2", channels_last=(layout == "hwc"), out=out)))
    nbytes = B * (S * S * 3 + So * So * 3)
    print(f"u8 cv2 {layout} {kind} B={B} {S}->{So} [{nbytes/1e6:.0f} MB]")
    for a in (1, 2, 4):
        print(f"   ahead={a}: " + "  ".join(f"R{R}c{c} {sorted(res[(a, R, c)])[1]*1e3:.1f}" for (aa, R, c) in variants if aa == a), flush=True)
