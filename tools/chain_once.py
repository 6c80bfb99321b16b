"""Dev tool: a few launches of every kernel of the main_batched chain (for rocprofv3 --pmc / --kernel-trace).
usage: chain_once.py [B S So]   (default 256 1024 500)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
B, S, So = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 1024, 500)
img8 = (torch.rand(B, S, S, 3, device=dev) * 255).to(torch.uint8)
m24 = torch.rand(B, 24, 24, device=dev)
for _ in range(6):
    pipeline.warp_from_masks(img8, m24, (So, So))
torch.cuda.synchronize()
