"""round 5: where does the unaligned uint8 resample lose?  684-wide images (aligned) on the aligned kernel, on the UA
instantiation (forced: tuning key bound=16), from a view that starts 1 byte in (every row misaligned by 1), and 683-wide."""
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
B, H, Ho, Wo = 256, 1024, 500, 500
g = torch.Generator(device=dev).manual_seed(1)
px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.5, 1); py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.5, 1)
out = torch.empty(B, Ho, Wo, 3, device=dev, dtype=torch.uint8)
for rep in range(3):
    line = []
    for W in (684, 683):
        flat = torch.randint(0, 256, (B * H * W * 3 + 16,), device=dev, dtype=torch.uint8, generator=g)
        mx, my = pipeline.axis_maps_from_pdf(px, py, (H, W), (Ho, Wo))
        for off in (0, 1, 2):
            img = flat[off:off + B * H * W * 3].view(B, H, W, 3)
            line.append(f"W={W} base+{off}: {t(lambda: cu.remap_separable(img, mx, my, mode='cv2', channels_last=True, out=out)):.4f}")
            if W == 684 and off == 0:
                with _lib.debug_override(bound=16):
                    line.append(f"W=684 base+0 UA forced: {t(lambda: cu.remap_separable(img, mx, my, mode='cv2', channels_last=True, out=out)):.4f}")
        del flat
    print("  ".join(line), flush=True)
