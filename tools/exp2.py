import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_bench import bench
for rep in range(2):
    bench(256, 1024, "hwc", "uniform", "exact")
    bench(256, 1024, "hwc", "uniform", "exact", tag="noload", remap_ldspad=1)
    bench(256, 1024, "hwc", "uniform", "exact", tag="nostore", remap_ldspad=2)
    bench(256, 1024, "hwc", "uniform", "exact", tag="neither", remap_ldspad=3)
    bench(256, 1024, "chw", "uniform", "exact")
    bench(256, 1024, "chw", "uniform", "exact", tag="noload", remap_ldspad=1)
    bench(256, 1024, "chw", "uniform", "exact", tag="nostore", remap_ldspad=2)
    bench(256, 1024, "chw", "uniform", "exact", tag="neither", remap_ldspad=3)
    bench(256, 1024, "hwc", "peaked", "exact")
    bench(256, 1024, "hwc", "peaked", "exact", tag="noload", remap_ldspad=1)
    bench(256, 1024, "hwc", "peaked", "exact", tag="nostore", remap_ldspad=2)
a = torch.rand(256, 1024, 1024, 3, device="cuda"); b = torch.empty_like(a)
for f, name, nb in ((lambda: torch.add(a, 1.0, out=b), "add", 2), (lambda: b.copy_(a), "copy", 2), (lambda: b.fill_(1.0), "fill", 1), (lambda: a.sum(), "sum", 1)):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[5]
    print(name, f"{ms:.4f} ms {nb*a.numel()*4/ms/1e9:.2f} TB/s")
