"""Dev tool: A/B two builds of libattwarp_hip.so on the float32 resample kernel (alternating subprocesses on the
same box).   usage: ab_remap.py libA.so libB.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from attwarp_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
    import remap_bench as rb
    tag = os.path.basename(os.environ["AB_LIB"])
    for rep in range(2):
        rb.bench(256, 1024, "chw", "uniform", "cv2", tag=tag)
        rb.bench(256, 1024, "chw", "uniform", "exact", tag=tag)
        rb.bench(64, 336, "hwc", "uniform", "cv2", 100, tag=tag)
        rb.bench(256, 336, "hwc", "uniform", "cv2", 50, tag=tag)
        rb.bench(256, 336, "hwc", "uniform", "exact", 50, tag=tag)
        rb.bench(256, 336, "chw", "uniform", "cv2", 50, tag=tag)
    rb.bench(256, 512, "hwc", "uniform", "cv2", 50, tag=tag)
    rb.bench(256, 336, "hwc", "peaked", "cv2", 50, tag=tag)
    rb.bench(256, 1024, "chw", "peaked", "cv2", tag=tag)
    rb.bench(256, 1024, "hwc", "uniform", "cv2", tag=tag)
else:
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    for rep in range(3):
        for lib in libs:
            subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, AB_LIB=lib))
