import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import checkpoint_utils as cu
from oracle import warp_oracle as O
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
H, W, C = 128, 128, 4
img = rng.random((1, H, W, C), dtype=np.float32)
mx = np.arange(W, dtype=np.float32)[None] * 0.999; my = np.arange(H, dtype=np.float32)[None] * 0.999
ref = O.remap_bilinear(img[0], mx[0], my[0])
for R in ("16", "1"):
    for alt in ("0", "1"):
        os.environ["ATTWARP_REMAP_ROWS"] = R; os.environ["ATTWARP_REMAP_ALT"] = alt
        out = torch.full((1, H, W, C), -7.0, device=dev)
        got = cu.remap_separable(torch.from_numpy(img).to(dev), torch.from_numpy(mx).to(dev), torch.from_numpy(my).to(dev), channels_last=True, out=out).cpu().numpy()[0]
        badrows = sorted(set(np.argwhere(got != ref)[:, 0].tolist()))
        print("R", R, "alt", alt, "bad rows", badrows[:40], "n", len(badrows))
        r = badrows[0] if badrows else 0
        g = got[r].reshape(-1); e = ref[r].reshape(-1)
        idx = np.nonzero(g != e)[0]
        print(" row", r, "bad elems", idx[:8], "...", idx[-3:], "got", g[idx[:4]], "exp", e[idx[:4]], "untouched(-7):", int((got == -7).sum()))
        for rr in range(H):
            if np.array_equal(got[r], ref[rr]): print("  got row", r, "== ref row", rr)
