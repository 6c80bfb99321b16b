"""Dev tool: run the remap kernel a few times on the BASELINE config-3 size (for rocprofv3 --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import checkpoint_utils as cu, pipeline
dev = torch.device("cuda:0")
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 1024)
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
g = torch.Generator(device=dev).manual_seed(1)
sc = 0.02 if kind == "uniform" else 2.0
px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * sc, 1)
py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * sc, 1)
mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
img = torch.rand((B, S, S, 3), device=dev); out = torch.empty_like(img)
from attwarp_amd import _lib
with _lib.debug_override(remap_noswz=int("noswz" in sys.argv)):
    for _ in range(5):
        cu.remap_separable(img, mx, my, channels_last=True, out=out)
torch.cuda.synchronize()
