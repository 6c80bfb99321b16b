"""Dev tool (round 4): the float32 staged resample at the bench sizes, both modes and map kinds, one line each
(`python tools/pair_ab.py [key=value ...]` forwards overrides to attwarp_debug_set)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import remap_bench as rb
over = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
for rep in range(2):
    for mode in ("cv2", "exact"):
        rb.bench(256, 1024, "hwc", "uniform", mode, 20, **over)
        rb.bench(256, 1024, "hwc", "peaked", mode, 20, **over)
        rb.bench(256, 1024, "chw", "uniform", mode, 20, **over)
        rb.bench(64, 336, "hwc", "uniform", mode, 50, **over)
        rb.bench(256, 336, "hwc", "uniform", mode, 50, **over)
        rb.bench(256, 336, "hwc", "peaked", mode, 50, **over)
        rb.bench(64, 768, "hwc", "uniform", mode, 50, **over)
        rb.bench(64, 2048, "hwc", "uniform", mode, 10, **over)
