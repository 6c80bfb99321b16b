import os, sys, torch, itertools
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import remap_bench as rb
import io, contextlib
res = {}
vars_ = [dict(remap_nt=8)] + [dict(remap_rows=R, remap_cpw=c, remap_noswz=g) for R in (2, 3, 4) for c in (1, 2, 3, 4) for g in (4, 8, 16)]
for rep in range(3):
    for over in vars_:
        for kind in ("uniform", "peaked"):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ms = rb.bench(256, 1024, "hwc", kind, "cv2", 10, **over)
            res.setdefault((tuple(sorted(over.items())), kind), []).append(ms)
for (k, kind), v in sorted(res.items(), key=lambda kv: (kv[0][1], min(kv[1]))):
    print(kind, dict(k), " ".join(f"{x:.4f}" for x in v), flush=True)
