"""Dev tool: calibrate the achievable copy bandwidth on this box (torch copy kernels)."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts)//2]
for nbytes in (3221225472, 805306368):
    a = torch.empty(nbytes // 4, device=dev, dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    ms = t(lambda: b.copy_(a))
    print(f"copy_ {nbytes/1e9:.2f} GB: {ms:.4f} ms  {2*nbytes/ms/1e9:.2f} TB/s (r+w)")
    ms = t(lambda: b.fill_(1.0))
    print(f"fill_ {nbytes/1e9:.2f} GB: {ms:.4f} ms  {nbytes/ms/1e9:.2f} TB/s (w)")
    ms = t(lambda: a.sum())
    print(f"sum   {nbytes/1e9:.2f} GB: {ms:.4f} ms  {nbytes/ms/1e9:.2f} TB/s (r)")
    ms = t(lambda: torch.add(a, 1.0, out=b))
    print(f"add   {nbytes/1e9:.2f} GB: {ms:.4f} ms  {2*nbytes/ms/1e9:.2f} TB/s (r+w)")
