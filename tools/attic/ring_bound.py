"""round 5: the three-slot LDS ring's UPPER BOUND on this lease (tuning key bound=1: only the top row staged, three-row LDS pool,
four workgroups per CU; garbage output) against the shipped float32 cv2 resample, alternating, plus torch.add."""
import os, sys, io, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import remap_bench as rb
res = {0: [], 1: []}
for rep in range(4):
    for b in (0, 1):
        with contextlib.redirect_stdout(io.StringIO()):
            res[b].append(rb.bench(256, 1024, "hwc", "uniform", "cv2", 10, **({"bound": 1} if b else {})))
dev = torch.device("cuda:0")
a = torch.rand(256, 1024, 1024, 3, device=dev); o = torch.empty_like(a)
for _ in range(3): torch.add(a, 1, out=o)
torch.cuda.synchronize(); ts = []
for _ in range(10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); torch.add(a, 1, out=o); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("shipped ms", [round(v, 4) for v in res[0]], " ring ceiling ms", [round(v, 4) for v in res[1]], f" torch.add {sorted(ts)[5]:.4f}", flush=True)
