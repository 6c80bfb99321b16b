"""Dev tool (round 4): the float32 staged resample with its look-ahead variants, same process, alternating.
remap_nt=8: the row loop of rounds 1-3 (register set of a look-ahead load chosen at run time: the compiler waits for the
data right behind the load); remap_nt=4: WRONG-PIXELS timing experiment with fixed alternating destinations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import remap_bench as rb
variants = [dict(remap_nt=8), dict(), dict(remap_pair=1)]
for rep in range(2):
    for over in variants:
        for mode in ("cv2", "exact"):
            rb.bench(256, 1024, "hwc", "uniform", mode, 20, **over)
            rb.bench(256, 1024, "hwc", "peaked", mode, 20, **over)
            rb.bench(256, 1024, "chw", "uniform", mode, 20, **over)
            rb.bench(64, 336, "hwc", "uniform", mode, 50, **over)
            rb.bench(256, 336, "hwc", "uniform", mode, 50, **over)
            rb.bench(256, 336, "hwc", "peaked", mode, 50, **over)
