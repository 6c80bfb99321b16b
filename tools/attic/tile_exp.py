"""round 5 experiment: 1024-float column tiles for the 1024x1024x3 float32 resample (tuning flavour: remap_tiled=2, remap_tile_ko=4)
against the shipped whole-row form, alternating in one process."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import remap_bench as rb
variants = [("shipped", {})] + [(f"tile4 R{r} swz{g}", dict(remap_tiled=2, remap_tile_ko=4, remap_rows=r, remap_noswz=g)) for r, g in ((6, 0), (8, 0), (12, 0), (16, 0), (8, 8), (24, 0), (32, 0))] + \
           [(f"tile8 R{r}", dict(remap_tiled=2, remap_tile_ko=8, remap_rows=r)) for r in (8, 16)]
for kind in ("uniform", "peaked"):
    res = {n: [] for n, _ in variants}
    for rep in range(3):
        for n, o in variants:
            with contextlib.redirect_stdout(io.StringIO()):
                res[n].append(rb.bench(256, 1024, "hwc", kind, "cv2", 10, **o))
    print(kind, " ".join(f"{n}={sorted(v)[1]:.4f}" for n, v in res.items()), flush=True)
