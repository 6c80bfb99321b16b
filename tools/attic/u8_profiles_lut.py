"""round 5: the byte-packed marginals kernel computes (double)byte + 1e-9 arithmetically for identity / square and through a
256-entry LDS table for sqrt / exp / log.  Which is faster?  (attention_axis_maps = marginals + finalize; the finalize is the
same for both.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import new_method as nm
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]
for B, S in ((256, 1024), (64, 336)):
    att = torch.randint(0, 256, (B, S, S), device=dev, dtype=torch.uint8)
    for rep in range(2):
        print(f"B={B} S={S}: " + "  ".join(f"{tr} {t(lambda: nm.attention_axis_maps(att, 500, 500, tr))*1e3:.1f} us" for tr in ("identity", "square", "sqrt", "log")), flush=True)
