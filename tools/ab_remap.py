"""Dev tool: A/B two builds of libattwarp_hip.so on the float32 resample kernel (alternating subprocesses).
usage: ab_remap.py libA.so libB.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from attwarp_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
    import remap_bench as rb
    print(os.path.basename(os.environ["AB_LIB"]), flush=True)
    rb.bench(256, 1024, "hwc", "uniform")
    rb.bench(256, 1024, "chw", "uniform")
    rb.bench(64, 336, "hwc", "uniform", 50)
else:
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    for rep in range(3):
        for lib in libs:
            subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, AB_LIB=lib))
