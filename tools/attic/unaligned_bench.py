"""round 5: do unaligned rows cost anything on the staged kernels?  uint8 (the main_batched chain's resample) and float32
images 683 wide (rows of 2049 bytes / 2049 floats: the portrait TextVQA case) against 684 wide, same process, alternating;
the generic gather kernel (where these shapes ran until round 4) beside them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import checkpoint_utils as cu, pipeline, _lib
dev = torch.device("cuda:0")

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]

for dt, B, Ho, Wo in ((torch.uint8, 256, 500, 500), (torch.float32, 128, 1024, 683), (torch.float32, 128, 1024, 684)):
    res = {}
    for rep in range(3):
        for W in (683, 684):
            H = 1024
            g = torch.Generator(device=dev).manual_seed(W)
            img = torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8, generator=g) if dt == torch.uint8 else torch.rand(B, H, W, 3, device=dev, generator=g)
            px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.5, 1); py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.5, 1)
            mx, my = pipeline.axis_maps_from_pdf(px, py, (H, W), (Ho, Wo if dt == torch.uint8 else W))
            out = torch.empty(B, Ho, mx.shape[1], 3, device=dev, dtype=dt)
            res.setdefault((W, "staged"), []).append(t(lambda: cu.remap_separable(img, mx, my, mode="cv2", channels_last=True, out=out)))
            if rep == 0:
                with _lib.debug_override(remap_variant=1):
                    res[(W, "gather")] = [t(lambda: cu.remap_separable(img, mx, my, mode="cv2", channels_last=True, out=out), 5)]
            gb = (img.numel() + out.numel()) * img.element_size() / 1e9
            del img, out
    s = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{str(dt)[6:]:8s} B={B} 1024 x W x 3 -> {Ho} x {'500' if dt == torch.uint8 else 'W'}: W=683 {s[(683, 'staged')]:.4f} ms  W=684 {s[(684, 'staged')]:.4f} ms  "
          f"ratio {s[(683, 'staged')] / s[(684, 'staged')]:.3f}   gather kernel: 683 {s[(683, 'gather')]:.4f} ms, 684 {s[(684, 'gather')]:.4f} ms   ({gb / s[(683, 'staged')]:.2f} TB/s staged at 683)", flush=True)
