"""Dev tool: block-order variants of the resample INSIDE the fused step (B=256 / B=64 336x336 ring), same process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from attwarp_amd import dist as D, pipeline, _lib
dev = torch.device("cuda:0")
for (B, K) in ((256, 24), (64, 48)):
    st = bench.Step(B, 336, dev, seed=5, mode="cv2", layout="hwc")
    for rep in range(2):
        for name, over in (("R4g4", dict(remap_rows=4, remap_noswz=4)), ("R6g0", dict(remap_rows=6, remap_noswz=0)), ("R4g0", dict(remap_rows=4, remap_noswz=0)),
                           ("R6g4", dict(remap_rows=6, remap_noswz=4)), ("R6g2", dict(remap_rows=6, remap_noswz=2)), ("R3g4", dict(remap_rows=3, remap_noswz=4))):
            with _lib.debug_override(**over):
                ow = pipeline.OverlappedWarp([x[0] for x in st.sets], [x[1] for x in st.sets], st.starts, channels_last=True, mode="cv2")
                wall, _ = bench.time_overlapped(ow, K, 2, D)
            print(f"B={B} {name}: {wall / K * 1e3:.4f} ms/step  {bench.step_bytes(B, 336) / (wall / K) / 1e12:.3f} TB/s", flush=True)
            del ow
