"""Dev tool: time the uint8 resample kernel (main_batched chain shapes) with a rows-per-block sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline, new_method as nm, checkpoint_utils as cu, _lib
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
Rs = sys.argv[1:] or ["auto"]
for (B, S, So, layout) in [(64, 336, 500, "hwc"), (256, 336, 500, "hwc"), (64, 1024, 500, "hwc"), (256, 1024, 1024, "hwc"),
                           (256, 336, 336, "chw"), (256, 1024, 1024, "chw")]:
    shape = (B, S, S, 3) if layout == "hwc" else (B, 3, S, S)
    img8 = (torch.rand(*shape, device=dev) * 255).to(torch.uint8)
    px = torch.softmax(torch.randn(B, 24, device=dev), 1)
    mx, my = pipeline.axis_maps_from_pdf(px, px, (S, S), (So, So))
    for R in Rs:
        for mode in ("exact", "cv2"):
            with _lib.debug_override(remap_rows=-1 if R == "auto" else int(R)):
                ms = t(lambda: cu.remap_separable(img8, mx, my, mode=mode, channels_last=(layout == "hwc")))
            print(f"u8 {layout} {mode} B={B} {S}->{So} R={R}: {ms*1e3:.1f} us  {(B*(S*S*3+So*So*3))/ms/1e9:.2f} TB/s")
