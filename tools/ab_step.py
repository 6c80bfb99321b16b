"""Dev tool: same-box A/B of library builds on the fused step (pipeline.OverlappedWarp over a ring of batches).
usage: ab_step.py libA.so libB.so ...   (alternating subprocesses, 2 rounds)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    from attwarp_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
    import torch, bench
    from attwarp_amd import dist as D, pipeline
    dev = torch.device("cuda:0")
    tag = os.path.basename(os.environ["AB_LIB"])
    for (B, S, K) in ((64, 336, 48), (256, 336, 24)):
        res, *_ = bench.small_workload(B, S, dev, 5, "cv2", "hwc", K, 2, D, torch, pipeline)
        print(f"{tag:12s} B={B} S={S}: fused {res['ms_per_step']:.4f} ms/step {res['step_TBps']:.3f} TB/s  same={res['bit_identical_to_serial']}  "
              f"eager {res['eager']['ms_per_step']:.4f} stages {res['eager']['stages_ms']}", flush=True)
        torch.cuda.empty_cache()
else:
    libs = [os.path.abspath(p) for p in sys.argv[1:]]
    for rep in range(2):
        for lib in libs:
            subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, AB_LIB=lib))
