"""Dev tool (round 4): which block order of the cv2 float32 resample wins on THIS lease -- row blocks per workgroup x XCD
group size at 1024x1024x3 B=256 (uniform maps), exact mode and torch.add beside them, alternating in one process.
One line per lease: run it on several (`gpurun -- python tools/lease_orders.py`) and compare the worst cases."""
import os, sys, io, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import remap_bench as rb
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
vars_ = [("cv2", dict(remap_rows=3, remap_cpw=c, remap_noswz=g)) for c, g in ((1, 4),)] + \
        [("cv2", dict(remap_rows=R, remap_cpw=c, remap_noswz=g, remap_cv2_double=1)) for R, c, g in ((3, 2, 8), (3, 3, 8), (3, 2, 4), (3, 2, 16), (4, 2, 8), (2, 3, 8))] + \
        [("exact", dict())]
res = {}
for rep in range(3):
    for mode, over in vars_:
        with contextlib.redirect_stdout(io.StringIO()):
            ms = rb.bench(256, 1024, "hwc", kind, mode, 10, **over)
        res.setdefault((mode + ("D" if over.get("remap_cv2_double") else ""), f"{over.get('remap_rows', '')}c{over.get('remap_cpw')}", over.get("remap_noswz")), []).append(ms)
dev = torch.device("cuda:0")
a = torch.rand(256, 1024, 1024, 3, device=dev); b = torch.empty_like(a)
for _ in range(3): torch.add(a, 1, out=b)
torch.cuda.synchronize(); ts = []
for _ in range(10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); torch.add(a, 1, out=b); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(" ".join(f"{m}{'' if g is None else f'_R{c}g{g}'}={sorted(v)[1]:.4f}" for (m, c, g), v in res.items()) + f" add={sorted(ts)[5]:.4f}", flush=True)
