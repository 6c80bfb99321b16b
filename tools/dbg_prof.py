import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import new_method as nm, _lib
from attwarp_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda:0")
for (h, w) in [(64, 260), (70, 132), (336, 336)]:
    rng = np.random.default_rng(h * 13 + w)
    att = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    att[1] = np.clip(rng.normal(128, 3, (h, w)), 0, 255).astype(np.uint8)
    att[2, : h // 2] = 0
    a = torch.from_numpy(att).to(dev)
    lib = _lib.load()
    B = 3
    nbytes = lib.attwarp_axis_sums_workspace_bytes(B, h, w)
    for tr in range(5):
        res = []
        for var in (-1, 1):
            ws = torch.zeros(nbytes // 8, dtype=torch.float64, device=dev)
            mx = torch.empty(B, w, device=dev); my = torch.empty(B, h, device=dev)
            with _lib.debug_override(profiles_variant=var):
                call("attwarp_axis_maps_from_attention", ptr(a), _lib.U8, B, h, w, w, h, tr, 1.0, 50.0, 0, ptr(mx), ptr(my), ptr(ws), stream_ptr(dev))
            torch.cuda.synchronize()
            res.append((ws.cpu().numpy().copy(), mx.cpu().numpy(), my.cpu().numpy()))
        (wa, mxa, mya), (wb, mxb, myb) = res
        col_a, col_b = wa[:B * w].reshape(B, w), wb[:B * w].reshape(B, w)
        nl = (len(wa) - B * w) // (B * h)
        ls_a, ls_b = wa[B * w:].reshape(B, h, nl), wb[B * w:].reshape(B, h, nl)
        print(h, w, "tr", tr, "col diff", np.argwhere(col_a != col_b)[:4].tolist(), "ls diff", np.argwhere(ls_a != ls_b)[:4].tolist(),
              "maps equal", np.array_equal(mxa, mxb), np.array_equal(mya, myb))
