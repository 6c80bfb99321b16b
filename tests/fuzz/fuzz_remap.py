"""Test infrastructure (imports oracle/): randomised differential run of the resample entry point against the oracle -- random batch / sizes /
channels / layout / dtype / mode / map kinds (monotone, wild, NaN / Inf / huge coordinates, identity), sizes picked so
that every kernel family is hit (staged rows, CHW plane split, column tiles, uint8 integer form, generic gather).
usage: fuzz_remap.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from attwarp_amd import checkpoint_utils as cu
from oracle import warp_oracle as O
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); n = 0; bad = 0; fams = {}
def dim():
    r = rng.random()
    if r < 0.5: return int(rng.integers(1, 70))
    if r < 0.8: return int(rng.integers(70, 400))
    if r < 0.95: return int(rng.choice([336, 500, 512, 1024, 1100]))
    return int(rng.integers(1100, 2300))
while time.time() - t0 < budget:
    B = int(rng.integers(1, 4)); C = int(rng.integers(1, 5))
    H, W, Ho, Wo = dim(), dim(), dim(), dim()
    if H * W * C > 3_000_000 or Ho * Wo * C > 3_000_000: continue
    layout = "hwc" if rng.random() < 0.5 else "chw"
    dt = np.float32 if rng.random() < 0.5 else np.uint8
    mode = "cv2" if rng.random() < 0.6 else "exact"
    kind = rng.choice(["mono", "wild", "special", "identity"])
    if kind == "mono":
        mx = np.sort(rng.random((B, Wo)) * (W + 2) - 1, axis=1).astype(np.float32)
        my = np.sort(rng.random((B, Ho)) * (H + 2) - 1, axis=1).astype(np.float32)
    elif kind == "wild":
        mx = (rng.random((B, Wo)) * (W + 6) - 3).astype(np.float32); my = (rng.random((B, Ho)) * (H + 6) - 3).astype(np.float32)
    elif kind == "identity":
        mx = np.tile(np.arange(Wo, dtype=np.float32) * (W / Wo), (B, 1)); my = np.tile(np.arange(Ho, dtype=np.float32) * (H / Ho), (B, 1))
    else:
        mx = (rng.random((B, Wo)) * W).astype(np.float32); my = (rng.random((B, Ho)) * H).astype(np.float32)
        sp = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, W - 1, W - 0.5, -0.5, 0.0, 65536.0, -65536.0, 1 / 64, W - 1 - 1 / 64], np.float32)
        for m in (mx, my):
            k = max(1, m.size // 8)
            m.reshape(-1)[rng.integers(0, m.size, k)] = rng.choice(sp, k)
    img = rng.random((B, H, W, C), dtype=np.float32)
    if dt == np.uint8: img = (img * 255).astype(np.uint8)
    with np.errstate(all="ignore"):
        ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
    xh = img if layout == "hwc" else np.ascontiguousarray(img.transpose(0, 3, 1, 2))
    view = rng.random()
    if view < 0.25:          # a view at an element offset into a larger allocation: base pointer not 16-byte aligned
        off = int(rng.integers(1, 8))
        flat = torch.zeros(xh.size + 8, dtype=torch.from_numpy(xh).dtype, device=dev)
        flat[off:off + xh.size] = torch.from_numpy(xh).to(dev).reshape(-1)
        x = flat[off:off + xh.size].view(xh.shape)
    elif view < 0.4:         # a non-contiguous view (every other image of a larger batch)
        big = torch.from_numpy(np.repeat(xh, 2, axis=0)).to(dev)
        x = big[::2]
    else:
        x = torch.from_numpy(xh).to(dev)
    tmx, tmy = torch.from_numpy(mx).to(dev), torch.from_numpy(my).to(dev)
    if rng.random() < 0.2:   # maps as column slices of a wider tensor (strided rows)
        tmx = torch.cat([tmx, tmx], dim=1)[:, :mx.shape[1]]
    got = cu.remap_separable(x, tmx, tmy, mode=mode, channels_last=(layout == "hwc")).cpu().numpy()
    if layout == "chw": got = got.transpose(0, 2, 3, 1)
    n += 1
    key = (layout, dt.__name__, mode, kind); fams[key] = fams.get(key, 0) + 1
    if not np.array_equal(got, ref, equal_nan=True):
        bad += 1
        print("MISMATCH", B, C, H, W, Ho, Wo, layout, dt.__name__, mode, kind, "max", np.nanmax(np.abs(got.astype(np.float64) - ref.astype(np.float64))), flush=True)
print(f"{n} cases, {bad} mismatches, {len(fams)} (layout, dtype, mode, map) families, {time.time() - t0:.0f} s")
