"""Dev tool: a few launches of the main_batched chain (warp_from_masks, B=64, 336 -> 500) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
img8 = (torch.rand(64, 336, 336, 3, device=dev) * 255).to(torch.uint8)
m24 = torch.rand(64, 24, 24, device=dev)
for _ in range(10):
    pipeline.warp_from_masks(img8, m24)
torch.cuda.synchronize()
