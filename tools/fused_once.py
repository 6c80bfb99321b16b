"""Dev tool: a few fused steps (pipeline.OverlappedWarp over a ring of batches, B=256 336x336) for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
B, S = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (256, 336)
st = bench.Step(B, S, dev, seed=5, mode="cv2", layout="hwc")
ow = pipeline.OverlappedWarp([x[0] for x in st.sets], [x[1] for x in st.sets], st.starts, channels_last=True, mode="cv2")
ow.prime(); ow.prime2()
for _ in range(12):
    ow.step()
torch.cuda.synchronize()
