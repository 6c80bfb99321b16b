"""Dev tool: a few attention_axis_maps launches (A13 maps from uint8 attention) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import new_method as nm
dev = torch.device("cuda:0")
for (B, S, So) in [(256, 1024, 500), (64, 336, 500), (1, 336, 500)]:
    au8 = (torch.rand(B, S, S, device=dev) * 255).to(torch.uint8)
    for _ in range(5):
        nm.attention_axis_maps(au8, So, So, "identity")
    torch.cuda.synchronize()
