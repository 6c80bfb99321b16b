import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_bench import bench
for rep in range(2):
    bench(256, 1024, "hwc", "uniform", "exact")
    bench(256, 1024, "hwc", "uniform", "exact", remap_ldspad=24576)
    bench(256, 1024, "hwc", "uniform", "cv2")
    bench(256, 1024, "hwc", "uniform", "cv2", remap_cv2_single=1)
    bench(256, 1024, "hwc", "uniform", "cv2", remap_cv2_single=1, remap_rows=8)
    bench(256, 1024, "hwc", "uniform", "cv2", remap_cv2_single=1, remap_rows=2)
    bench(256, 1024, "chw", "uniform", "cv2")
    bench(256, 1024, "chw", "uniform", "cv2", remap_chw_split=0)
    bench(256, 1024, "chw", "uniform", "exact", remap_rows=4)
    bench(256, 1024, "chw", "uniform", "exact", remap_rows=8)
    bench(256, 1024, "chw", "uniform", "exact", remap_rows=12)
