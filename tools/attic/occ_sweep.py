"""Dev tool: workgroups per CU of the float32 resample (extra LDS through the test hook) x rows per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from remap_bench import bench
for mode in ("cv2", "exact"):
    bench(256, 1024, "hwc", "uniform", mode)
    for pad in (0, 8000, 16000, 30000):
        for R in (4, 8, 16):
            bench(256, 1024, "hwc", "uniform", mode, remap_ldspad=pad, remap_rows=R)
    bench(256, 1024, "hwc", "uniform", mode)
