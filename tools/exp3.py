import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_bench import bench
for rep in range(1):
    bench(256, 1024, "hwc", "uniform", "exact")
    for R in (4, 8, 16, 32):
        bench(256, 1024, "hwc", "uniform", "exact", tag="neither", remap_ldspad=3, remap_rows=R)
    bench(256, 1024, "hwc", "uniform", "exact", tag="neither,nogatherlerp", remap_ldspad=3+4)
    bench(256, 1024, "hwc", "uniform", "exact", tag="neither,noblend", remap_ldspad=3+8)
    bench(256, 1024, "hwc", "uniform", "exact", tag="neither,noboth", remap_ldspad=3+12)
    bench(256, 1024, "hwc", "uniform", "exact", tag="prologue only", remap_ldspad=3+16)
    bench(256, 1024, "hwc", "uniform", "exact", tag="prologue only R8", remap_ldspad=3+16, remap_rows=8)
    bench(256, 1024, "hwc", "uniform", "exact", tag="full,nogatherlerp", remap_ldspad=4)
    bench(256, 1024, "hwc", "uniform", "exact", tag="full,noblend", remap_ldspad=8)
    bench(256, 1024, "hwc", "uniform", "exact", tag="full,noboth", remap_ldspad=12)
    bench(256, 1024, "hwc", "uniform", "exact", tag="noload", remap_ldspad=1, remap_rows=8)
    bench(256, 1024, "hwc", "uniform", "exact", tag="nostore", remap_ldspad=2, remap_rows=8)
    bench(256, 1024, "hwc", "uniform", "exact", tag="noload", remap_ldspad=1, remap_rows=16)
    bench(256, 1024, "hwc", "uniform", "exact", tag="nostore", remap_ldspad=2, remap_rows=16)
