"""Dev tool: rows per workgroup of the float32 resample at 336x336 (wave quantisation: workgroups vs resident slots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from remap_bench import bench
for B in (64, 256):
    bench(B, 336, "hwc", "uniform", "cv2", 60)
    for R in range(4, 17):
        bench(B, 336, "hwc", "uniform", "cv2", 60, remap_rows=R)
    bench(B, 336, "hwc", "uniform", "cv2", 60)
