"""Dev tool: a few clip_preprocess launches (B=64, 500x500x3 -> 336) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
w500 = (torch.rand(64, 500, 500, 3, device=dev) * 255).to(torch.uint8)
for _ in range(10):
    pipeline.clip_preprocess(w500)
torch.cuda.synchronize()
