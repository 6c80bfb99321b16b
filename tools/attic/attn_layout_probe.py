"""Dev tool: does the attention reduce depend on the slice layout (alignment of the 576-token window, gap between rows)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
x = torch.empty(1 << 28, device=dev)
for _ in range(200): x.add_(1.0)
B = 256
for kv, s0, spread in ((640, 35, 8), (640, 32, 1), (640, 36, 1), (640, 33, 1), (640, 34, 1), (640, 35, 1), (640, 32, 8), (576, 0, 1)):
    rows = torch.softmax(torch.randn(20, B, 32, kv, device=dev), -1)
    starts = (s0 + torch.arange(B, device=dev) % spread).int()
    st = starts.repeat(20)
    for name, r in (("fp32", rows), ("fp16", rows.half())):
        for cyc in range(2):
            ms = timeit(lambda: pipeline.attention_step_maps(r, starts, 576, st))
        nb = 20 * B * 32 * 576 * r.element_size()
        print(f"kv={kv} start={s0}+i%{spread} {name}: {ms*1e3:.1f} us  {nb/ms/1e9:.2f} TB/s", flush=True)
