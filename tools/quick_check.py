"""Scratch GPU check: parity of the core kernels vs the oracle + remap timing (dev tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import checkpoint_utils as cu, _lib
from oracle import warp_oracle as O

dev = torch.device("cuda:0")
torch.manual_seed(0)
rng = np.random.default_rng(0)

def t(x): return torch.from_numpy(np.ascontiguousarray(x)).to(dev)

# --- per-axis chain
y = torch.softmax(torch.randn(4, 24) * 2, 1).numpy()
for L in (336, 500, 1024):
    x_o = O.upsample_pdf_right_inverse(y, L)
    x_g = cu.upsample_pdf_right_inverse(t(y), L).cpu().numpy()
    F_o = O.cdf_from_density(np.maximum(x_o, 0))
    F_g = cu.cdf_from_density(t(np.maximum(x_o, 0))).cpu().numpy()
    mx_o, my_o = O.maps_from_cdf(F_o, F_o, (L, L))
    mx_g, my_g = cu.axis_maps_from_cdf(t(F_o), t(F_o), (L, L))
    print(L, "rinv", np.array_equal(x_o, x_g), np.abs(x_o - x_g).max(), "cdf", np.array_equal(F_o, F_g),
          "map", np.array_equal(mx_o, mx_g.cpu().numpy()), np.abs(mx_o - mx_g.cpu().numpy()).max())

# --- remap parity
for (H, W, Ho, Wo, C) in [(336, 336, 336, 336, 3), (64, 96, 50, 70, 3), (33, 47, 40, 31, 1), (128, 128, 128, 128, 4)]:
    for dt in (np.float32, np.uint8):
        img = rng.random((2, H, W, C), dtype=np.float32)
        if dt == np.uint8: img = (img * 255).astype(np.uint8)
        px = torch.softmax(torch.randn(2, 24) * 2, 1).numpy(); py = torch.softmax(torch.randn(2, 24) * 2, 1).numpy()
        Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, W), 0))
        Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, H), 0))
        mx, my = O.maps_from_cdf(Fx, Fy, (Ho, Wo))
        ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b]) for b in range(2)])
        for var in ("r", "g"):
            os.environ["ATTWARP_REMAP_VARIANT"] = var
            o_hwc = cu.remap_separable(t(img), t(mx), t(my), channels_last=True).cpu().numpy()
            o_chw = cu.remap_separable(t(img.transpose(0, 3, 1, 2)), t(mx), t(my)).cpu().numpy().transpose(0, 2, 3, 1)
            d1 = np.abs(o_hwc.astype(np.float64) - ref).max(); d2 = np.abs(o_chw.astype(np.float64) - ref).max()
            print((H, W, Ho, Wo, C), dt.__name__, var, "hwc", d1, "chw", d2)

# --- timing
def bench(B, S, layout, var, iters=10):
    os.environ["ATTWARP_REMAP_VARIANT"] = var
    img = torch.rand((B, S, S, 3) if layout == "hwc" else (B, 3, S, S), device=dev)
    px = torch.softmax(torch.randn(B, 24, device=dev) * 2, 1)
    Fx = cu.cdf_from_density(cu.upsample_pdf_right_inverse(px, S).clamp_min(0))
    mx, my = cu.axis_maps_from_cdf(Fx, Fx, (S, S))
    for _ in range(3): out = cu.remap_separable(img, mx, my, channels_last=(layout == "hwc"))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): out = cu.remap_separable(img, mx, my, channels_last=(layout == "hwc"))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gb = 2 * B * S * S * 3 * 4 / 1e9
    print(f"B={B} S={S} {layout} var={var} R={os.environ.get('ATTWARP_REMAP_ROWS','auto')}: {ms:.3f} ms  {gb/ms*1e3/1e3:.2f} TB/s  {B/ms*1e3:.0f} img/s")

for layout in ("hwc", "chw"):
    for var in ("r", "g"):
        bench(64, 336, layout, var)
        bench(256, 1024, layout, var)
for R in (8, 16, 32, 64):
    os.environ["ATTWARP_REMAP_ROWS"] = str(R)
    bench(256, 1024, "hwc", "r")
    bench(256, 336, "hwc", "r")
