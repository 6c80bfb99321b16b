"""Dev tool: a few MarginalNet(1024, 4096, 256) inference forwards at B=256 (BASELINE configs[4]) for rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import model
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = model.MarginalNet(1024, 4096, 256).to(dev).eval()
fmap = torch.randn(B, 1024, 24, 24, device=dev)
tok = torch.randn(B, 32, 4096, device=dev); msk = (torch.rand(B, 32, 1, device=dev) > 0.3).float()
with torch.no_grad():
    for _ in range(6):
        net(fmap, 24, 24, tok, msk)
torch.cuda.synchronize()
