import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import attention_extraction as ae, _lib
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for (B, S) in [(256, 1024), (64, 336), (256, 336), (256, 512)]:
    rev = torch.rand(B, 24, 24, device=dev)
    ref = None
    for R in (-1, 16, 32, 64, 128, 256, 512):
        with _lib.debug_override(lanczos_rows=R):
            ms = timeit(lambda: ae.upsample_mask_lanczos(rev, (S, S)))
            out = ae.upsample_mask_lanczos(rev, (S, S))
        if ref is None: ref = out
        print(f"B={B} S={S} R={R}: {ms*1e3:.1f} us  {B*S*S/ms/1e6:.0f} GB/s  same={bool(torch.equal(out, ref))}", flush=True)
