"""Dev tool: a few attention-reduce launches (fp32 and fp16, bench shape) for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
B = 256
rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
starts = (35 + torch.arange(B, device=dev) % 8).int()
st = starts.repeat(20)
r16 = rows.half()
for _ in range(5):
    pipeline.attention_step_maps(rows, starts, 576, st)
    pipeline.attention_step_maps(r16, starts, 576, st)
torch.cuda.synchronize()
