import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import new_method as nm, _lib
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for (B, S) in [(256, 1024), (64, 336)]:
    mota = torch.randint(0, 256, (B, S, S), dtype=torch.uint8, device=dev)
    for stop in (1, 2, 3, 4, 5, -1):
        with _lib.debug_override(lanczos_rows=stop):
            ms = timeit(lambda: nm.attention_axis_maps(mota, 500, 500, "identity"))
        print(B, S, "stop", stop, f"{ms*1e3:.1f} us", flush=True)
